// hadoop_streams.hip -- batched Hadoop LZ4 / Snappy block streams for gfx950 (second half).
//
// The format: a sequence of [BE int plaintext bytes of the block][BE int compressed bytes][codec block] ... where a "block" may come
// as several [compressed bytes][codec block] chunks (the reference's writer always emits one).  Replaces, over the HIP block codecs,
//   the writers   M/lz4/Lz4HadoopOutputStream.java:60-118, M/snappy/SnappyHadoopOutputStream.java:60-131 used as "write everything, close"
//                 (T/HadoopCodecCompressor.java:57-72): chunks of bufferSize - overhead plaintext bytes, overhead = max((int)(size * 0.01), 10)
//                 for LZ4 and size / 6 + 32 for Snappy; bufferSize 256 KiB unless configured (context option hadoop.buffer_size);
//   the readers   M/lz4/Lz4HadoopInputStream.java:47-156, M/snappy/SnappyHadoopInputStream.java:44-170 read to the end the way the
//                 reference's test harness does (T/HadoopCodecDecompressor.java:40-60): read(output, done, capacity - done) until -1 or
//                 full, then one read() -- a byte there is "All input was not consumed" (ACHIP_D_HDP_NOT_CONSUMED).
// An item of the batch is a whole stream.
//
// Reader, the fast way (chunk-parallel, as for x-snappy-framed streams in snappy_frame.hip):
//   walk    a LANE per stream runs over the length fields only.  It accepts the shape every writer produces -- each block one chunk, the
//           chunk producing exactly the block's declared length (Snappy announces it in its preamble; for LZ4 it is assumed and checked
//           afterwards), every block fitting what the destination has left (so the Java reader decodes straight into the caller's buffer,
//           with the remaining capacity) -- and appends one descriptor per chunk to a batch that exists on the device only;
//   decode  that batch runs through the batched LZ4 / Snappy block decoders, chosen on the device as for any batch;
//   fold    a lane per stream: all chunks decoded to the declared lengths -> done; anything else -> the serial kernel.
// Reader, the general way (and the fallback): a WAVEFRONT per stream runs the Java loops themselves -- same checks in the same order,
// the stream's own buffer included (LZ4: bufferSize + 8 bytes; Snappy: grown to the largest chunk + 8), because its capacity decides
// what the block decoder says about a malformed chunk -- with the 64-lane ring block decoders inside each step.
//
// Writer: a chunk's position depends on the sizes before it, but every chunk except a stream's last is full and none is larger than
// the codec's bound, so chunk k is compressed at its WORST-CASE position k * (8 + maxCompressedLength(chunk)) by persistent wavefronts
// drawing chunks from one list; a wavefront per stream then writes the length fields and moves the chunks left into place, in order.
//
// What the one-shot form adds: the offset of the stream-level IOExceptions (the position where the failing read began); for the
// Snappy reader a negative chunk length other than -1 is ACHIP_D_HDP_NEGATIVE_LENGTH (Java goes on with whatever its buffer holds from the
// chunk before: an exception whose kind depends on stale state; the LZ4 reader takes any negative length for the end of the stream, as
// Java does); a Snappy chunk that ends inside its length
// preamble is ACHIP_D_SNAPPY_TRUNCATED (Java reads stale buffer bytes); a Snappy chunk that announces more than the destination has
// left AND more than the wavefront's buffer holds (max(bufferSize, 256 KiB) + 8) is not decoded to look for errors in it
// (ACHIP_D_HDP_NOT_CONSUMED, what Java reports for a well-formed one).
#include "lz4_decode_body.h"
#include "snappy_decode_body.h"
#include "lz4_compress_mw.h"
#include "snappy_compress_mw.h"

namespace achip {

namespace hdp {
constexpr int IN_RING = 2048, OUT_RING = 4096;
constexpr int32_t STREAM_EOF = -1000000;  // internal: end of stream
constexpr int32_t MAX_CHUNKS = 1 << 20;

__host__ __device__ __forceinline__ int32_t input_max_size(bool snappy, int32_t bufferSize)
{
    const int32_t lz4Overhead = (int32_t)(bufferSize * 0.01) > 10 ? (int32_t)(bufferSize * 0.01) : 10;
    return bufferSize - (snappy ? bufferSize / 6 + 32 : lz4Overhead);
}
__host__ __device__ __forceinline__ int64_t block_bound(bool snappy, int64_t n)
{
    return snappy ? 32 + n + n / 6 : n + n / 255 + 16;  // M/snappy/SnappyRawCompressor.java:69 ; M/lz4/Lz4RawCompressor.java:64-67
}

__device__ __forceinline__ int32_t rd_be(const uint8_t* p)
{
    return (int32_t)(((uint32_t)p[0] << 24) + ((uint32_t)p[1] << 16) + ((uint32_t)p[2] << 8) + (uint32_t)p[3]);
}
__device__ __forceinline__ void wr_be(uint8_t* p, int32_t v)
{
    p[0] = (uint8_t)((uint32_t)v >> 24);
    p[1] = (uint8_t)((uint32_t)v >> 16);
    p[2] = (uint8_t)((uint32_t)v >> 8);
    p[3] = (uint8_t)v;
}

// readUncompressedLength (M/snappy/SnappyRawDecompressor.java:277-321) of a chunk's data; returns the length or a status
__device__ __forceinline__ int32_t snappy_announced(const uint8_t* in, int32_t len, int32_t& eoOut)
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
    return (int32_t)expected;
}

// ---------------------------------------------------------------------------------------------------------------------
// the general reader: one wavefront per stream (everything wave-uniform)
struct Reader {
    const uint8_t* in;
    int32_t n, pos;
    int32_t blockLen, chunkOff, chunkLen;  // uncompressedBlockLength, uncompressedChunkOffset / Length
    uint8_t* internal;                     // the stream's own buffer (`uncompressedChunk`)
    int32_t internalLen;                   // its Java length
    int32_t internalMax;                   // what this wavefront can hold
    const uint8_t* chunk;                  // `compressed`
    int64_t eo;
    uint8_t* lds;
    int lane;
    bool snappy;
};

// readBigEndianInt :142-156
__device__ __forceinline__ int32_t read_be(Reader& r, int32_t& v)
{
    if (r.pos >= r.n) {
        return STREAM_EOF;
    }
    if (r.n - r.pos < 4) {
        r.eo = r.pos;
        return mk_status(ACHIP_CLASS_MALFORMED, ACHIP_D_HDP_TRUNCATED_INT);
    }
    v = rd_be(r.in + r.pos);
    r.pos += 4;
    return v == -1 ? STREAM_EOF : 0;
}

// the common head of bufferCompressedData (Lz4HadoopInputStream.java:100-127) and readNextChunk (SnappyHadoopInputStream.java:91-113);
// returns the chunk's compressed length, STREAM_EOF or a status
__device__ __forceinline__ int32_t next_chunk(Reader& r)
{
    r.blockLen -= r.chunkOff;
    r.chunkOff = 0;
    r.chunkLen = 0;
    while (r.blockLen == 0) {
        int32_t v = 0;
        const int32_t e = read_be(r, v);
        if (e == STREAM_EOF) {
            r.blockLen = 0;
            return STREAM_EOF;
        }
        if (e < 0) {
            return e;
        }
        r.blockLen = v;
    }
    int32_t clen = 0;
    const int32_t e = read_be(r, clen);
    if (e != 0) {
        return e;
    }
    if (clen < 0) {
        if (!r.snappy) {  // Lz4HadoopInputStream.java:51-54,65-68: `compressedChunkLength < 0` -- any negative value ends the stream for this read
            return STREAM_EOF;
        }
        r.eo = r.pos - 4;
        return mk_status(ACHIP_CLASS_MALFORMED, ACHIP_D_HDP_NEGATIVE_LENGTH);
    }
    if (clen > r.n - r.pos) {  // readInput :129-140
        r.eo = r.pos;
        return mk_status(ACHIP_CLASS_MALFORMED, ACHIP_D_HDP_EOF_BLOCK_DATA);
    }
    r.chunk = r.in + r.pos;
    r.pos += clen;
    return clen;
}

// the LZ4 block decoder as Lz4JavaDecompressor.decompress(byte[], ...) : returns the length or a status (eo = the codec's offset)
__device__ __forceinline__ int32_t lz4_decode(Reader& r, const uint8_t* src, int32_t len, uint8_t* dst, int32_t cap)
{
    using FR = Rings<64, IN_RING, OUT_RING, 1>;
    FR R;
    R.init(r.lds, r.lds + IN_RING, src, len, dst, r.lane);
    int32_t bst = 0, beo = 0, bop = 0;
    wave_mem_order();
    lz4_block_decode<64, IN_RING, OUT_RING, 1>(R, src, len, cap, bst, beo, bop);
    wave_mem_order();
    if (bst != 0) {
        r.eo = beo;
        return bst;
    }
    return bop;
}
__device__ __forceinline__ int32_t snappy_decode(Reader& r, const uint8_t* src, int32_t len, uint8_t* dst, int32_t cap)
{
    int32_t bst = 0, beo = 0, bop = 0;
    wave_mem_order();
    snappy_buffer_decode<64, IN_RING, OUT_RING, 1>(r.lds, r.lds + IN_RING, nullptr, src, len, dst, cap, r.lane, bst, beo, bop);
    wave_mem_order();
    if (bst != 0) {
        r.eo = beo;
        return bst;
    }
    return bop;
}

// Lz4HadoopInputStream.read(byte[], int, int) :61-82 ; single = read() :47-58 (returns 0 for "a byte")
__device__ int32_t lz4_read(Reader& r, uint8_t* dst, int32_t length, bool single)
{
    while (r.chunkOff >= r.chunkLen) {
        const int32_t clen = next_chunk(r);
        if (clen < 0) {
            return clen;
        }
        if (!single && length >= r.blockLen) {  // favor writing directly to the user buffer
            const int32_t w = lz4_decode(r, r.chunk, clen, dst, length);
            if (w < 0) {
                return w;
            }
            r.chunkLen = w;
            r.chunkOff = w;
            return w;
        }
        const int32_t w = lz4_decode(r, r.chunk, clen, r.internal, r.internalLen);
        if (w < 0) {
            return w;
        }
        r.chunkLen = w;
    }
    if (single) {
        r.chunkOff++;
        return 0;
    }
    const int32_t size = length < r.chunkLen - r.chunkOff ? length : r.chunkLen - r.chunkOff;
    wave_mem_order();
    group_copy<64>(dst, r.internal + r.chunkOff, size, r.lane);
    wave_mem_order();
    r.chunkOff += size;
    return size;
}

// SnappyHadoopInputStream.read(byte[], int, int) :57-73 over readNextChunk :91-141 ; single = read() :44-54
__device__ int32_t snappy_read(Reader& r, uint8_t* dst, int32_t length, bool single)
{
    if (r.chunkOff >= r.chunkLen) {
        bool direct = false;
        const int32_t clen = next_chunk(r);
        if (clen != STREAM_EOF) {
            if (clen < 0) {
                return clen;
            }
            int32_t beo = 0;
            const int32_t announced = snappy_announced(r.chunk, clen, beo);
            if (announced < 0) {
                r.eo = beo;
                return announced;
            }
            r.chunkLen = announced;
            if (r.chunkLen > r.blockLen) {
                r.eo = r.pos - clen;
                return mk_status(ACHIP_CLASS_MALFORMED, ACHIP_D_HDP_CHUNK_EXCEEDS_BLOCK);
            }
            direct = !single;
            uint8_t* target = single ? r.internal : dst;
            int32_t cap = single ? r.internalLen : length;
            if (r.chunkLen > cap) {
                if (r.internalLen < r.chunkLen) {
                    if (r.chunkLen > r.internalMax - 8) {  // (beyond this wavefront's buffer: see the file header)
                        r.eo = r.pos;
                        return mk_status(ACHIP_CLASS_OUTPUT_TOO_SMALL, ACHIP_D_HDP_NOT_CONSUMED);
                    }
                    r.internalLen = r.chunkLen + 8;
                }
                direct = false;
                target = r.internal;
                cap = r.internalLen;
            }
            const int32_t w = snappy_decode(r, r.chunk, clen, target, cap);
            if (w < 0) {
                return w;
            }
            if (w != r.chunkLen) {
                r.eo = r.pos - clen;
                return mk_status(ACHIP_CLASS_MALFORMED, ACHIP_D_HDP_LENGTH_MISMATCH);
            }
        }
        if (r.chunkLen == 0) {
            return STREAM_EOF;
        }
        if (direct) {
            r.chunkOff += r.chunkLen;
            return r.chunkLen;
        }
    }
    if (single) {
        r.chunkOff++;
        return 0;
    }
    const int32_t size = length < r.chunkLen - r.chunkOff ? length : r.chunkLen - r.chunkOff;
    wave_mem_order();
    group_copy<64>(dst, r.internal + r.chunkOff, size, r.lane);
    wave_mem_order();
    r.chunkOff += size;
    return size;
}

// T/HadoopCodecDecompressor.java:40-60
template <bool SNAPPY>
__device__ int32_t decompress_item(const uint8_t* in, int32_t inLen, uint8_t* out, int32_t outCap, uint8_t* internal, int32_t internalMax, int32_t bufferSize, uint8_t* lds, int lane,
                                   int32_t& doneOut, int64_t& eo)
{
    Reader r;
    r.in = in;
    r.n = inLen;
    r.pos = 0;
    r.blockLen = 0;
    r.chunkOff = 0;
    r.chunkLen = 0;
    r.internal = internal;
    r.internalLen = SNAPPY ? 0 : bufferSize + 8;
    r.internalMax = internalMax;
    r.chunk = in;
    r.eo = 0;
    r.lds = lds;
    r.lane = lane;
    r.snappy = SNAPPY;
    int32_t done = 0;
    int32_t result = 0;
    while (done < outCap) {
        const int32_t size = SNAPPY ? snappy_read(r, out + done, outCap - done, false) : lz4_read(r, out + done, outCap - done, false);
        if (size == STREAM_EOF) {
            break;
        }
        if (size < 0) {
            result = size;
            break;
        }
        done += size;
    }
    if (result == 0) {
        const int32_t b = SNAPPY ? snappy_read(r, nullptr, 0, true) : lz4_read(r, nullptr, 0, true);
        if (b >= 0) {
            r.eo = r.pos;
            result = mk_status(ACHIP_CLASS_OUTPUT_TOO_SMALL, ACHIP_D_HDP_NOT_CONSUMED);
        }
        else if (b != STREAM_EOF) {
            result = b;
        }
    }
    doneOut = done;
    eo = r.eo;
    return result;
}

template <bool SNAPPY>
__global__ __launch_bounds__(64) void hadoop_serial_decompress_kernel(BatchArgs a, uint8_t* internals, int32_t internalMax, int32_t bufferSize, int32_t* nextItem, const int32_t* only)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[IN_RING + OUT_RING];
    __shared__ int32_t item;
    const int lane = threadIdx.x;
    uint8_t* internal = internals + (size_t)blockIdx.x * (size_t)internalMax;
    for (;;) {
        __syncthreads();
        if (lane == 0) {
            item = atomicAdd(nextItem, 1);
        }
        __syncthreads();
        const int32_t s = item;
        if (s >= a.nBlocks) {
            return;
        }
        if (only != nullptr && only[s] == 0) {
            continue;  // done by the chunk-parallel path
        }
        int32_t done = 0;
        int64_t eo = 0;
        const int32_t st = decompress_item<SNAPPY>(a.srcBase + a.srcOff[s], a.srcLen[s], a.dstBase + a.dstOff[s], a.dstCap[s], internal, internalMax, bufferSize, lds, lane, done, eo);
        if (lane == 0) {
            a.outLen[s] = st == 0 ? done : 0;
            a.status[s] = st;
            a.errOffset[s] = st == 0 ? 0 : eo;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// the chunk-parallel reader
struct ChunkList {
    // per stream
    int32_t* sFirst;
    int32_t* sCount;
    int32_t* sOut;      // plaintext bytes if every chunk decodes as declared
    int32_t* sSerial;   // 1: the serial kernel takes the stream
    // per chunk: a batch for the block decoders ...
    int64_t* cSrcOff;
    int32_t* cSrcLen;
    int64_t* cDstOff;
    int32_t* cDstCap;
    int32_t* cOutLen;
    int32_t* cStatus;
    int64_t* cErrOff;
    int32_t* cExpect;   // ... and the block length it has to produce
    int32_t* counters;  // [0] chunks allocated, [1] chunks in the batch, [16..] probe statistics of the decoders' auto choice
};

// The Java loops over one stream without the chunk bodies; returns false when the stream is not of the simple shape (the serial kernel
// then decodes it).  FILL = false: count the chunks; true: write their descriptors.
template <bool SNAPPY, bool FILL>
__device__ bool walk_stream(const BatchArgs& a, const ChunkList& L, int32_t stream, int32_t first, int32_t& countOut, int32_t& outOut)
{
    const uint8_t* __restrict__ in = a.srcBase + a.srcOff[stream];
    const int32_t inLen = a.srcLen[stream];
    const int32_t outCap = a.dstCap[stream];
    int32_t pos = 0, o = 0, n = 0;
    countOut = 0;
    outOut = 0;
    while (pos < inLen) {
        if (inLen - pos < 4) {
            return false;
        }
        const int32_t u = rd_be(in + pos);
        pos += 4;
        if (u == 0) {
            continue;  // `while (uncompressedBlockLength == 0)`
        }
        if (u < 0 || inLen - pos < 4) {
            return false;
        }
        const int32_t clen = rd_be(in + pos);
        pos += 4;
        if (clen < 0 || clen > inLen - pos || u > outCap - o) {
            return false;
        }
        if (SNAPPY) {
            int32_t beo = 0;
            if (snappy_announced(in + pos, clen, beo) != u) {
                return false;
            }
        }
        if (FILL) {
            const int32_t c = first + n;
            L.cSrcOff[c] = a.srcOff[stream] + pos;
            L.cSrcLen[c] = clen;
            L.cDstOff[c] = a.dstOff[stream] + o;
            L.cDstCap[c] = outCap - o;  // (what the Java reader hands its block decoder: the rest of the caller's buffer)
            L.cExpect[c] = u;
        }
        n++;
        o += u;
        pos += clen;
    }
    countOut = n;
    outOut = o;
    return true;
}

template <bool SNAPPY>
__global__ __launch_bounds__(64) void hadoop_walk_kernel(BatchArgs a, ChunkList L)
{
    const int32_t stream = blockIdx.x * 64 + threadIdx.x;
    if (stream >= a.nBlocks) {
        return;
    }
    int32_t n = 0, out = 0;
    const bool simple = walk_stream<SNAPPY, false>(a, L, stream, 0, n, out);
    const int32_t first = simple && n > 0 ? atomicAdd(L.counters, n) : 0;
    const bool fits = simple && (int64_t)first + n <= MAX_CHUNKS;
    L.sFirst[stream] = first;
    L.sCount[stream] = fits ? n : 0;
    L.sOut[stream] = out;
    L.sSerial[stream] = fits ? 0 : 1;
    if (fits && n > 0) {
        walk_stream<SNAPPY, true>(a, L, stream, first, n, out);
    }
    else if (simple && !fits) {  // the part of this stream's range that lies inside the arrays: empty blocks nobody looks at
        for (int64_t c = first; c < (int64_t)first + n && c < MAX_CHUNKS; c++) {
            L.cSrcOff[c] = 0;
            L.cSrcLen[c] = 0;
            L.cDstOff[c] = 0;
            L.cDstCap[c] = 0;
            L.cExpect[c] = -1;
        }
    }
}

__global__ void hadoop_seal_kernel(ChunkList L)
{
    const int32_t allocated = L.counters[0];
    L.counters[1] = allocated < MAX_CHUNKS ? allocated : MAX_CHUNKS;
}

__global__ __launch_bounds__(64) void hadoop_fold_kernel(BatchArgs a, ChunkList L)
{
    const int32_t stream = blockIdx.x * 64 + threadIdx.x;
    if (stream >= a.nBlocks || L.sSerial[stream] != 0) {
        return;
    }
    const int32_t first = L.sFirst[stream], n = L.sCount[stream];
    bool ok = true;
    for (int32_t k = 0; k < n; k++) {
        ok = ok && L.cStatus[first + k] == 0 && L.cOutLen[first + k] == L.cExpect[first + k];
    }
    if (ok) {
        a.outLen[stream] = L.sOut[stream];
        a.status[stream] = 0;
        a.errOffset[stream] = 0;
    }
    else {
        L.sSerial[stream] = 1;  // a chunk failed or did not produce its block's length: the Java loops decide what that means
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// the writer
struct BlockList {
    int32_t* sFirst;   // per stream
    int32_t* sCount;
    int32_t* sStatus;
    int32_t* bStream;  // per chunk
    int32_t* bIndex;
    int32_t* bSize;    // compressed bytes at the worst-case position
    int32_t* counters; // [0] chunks allocated, [1] chunks in the list, [2] encode cursor, [3] compact cursor
};

template <bool SNAPPY>
__global__ __launch_bounds__(64) void hadoop_plan_kernel(BatchArgs a, BlockList L, int32_t bufferSize)
{
    const int32_t stream = blockIdx.x * 64 + threadIdx.x;
    if (stream >= a.nBlocks) {
        return;
    }
    const int32_t inLen = a.srcLen[stream];
    const int32_t chunk = input_max_size(SNAPPY, bufferSize);
    int32_t st = 0;
    int64_t chunks = 0;
    if (inLen < 0 || chunk <= 0) {
        st = mk_status(ACHIP_CLASS_INVALID_ARGUMENT, ACHIP_D_BAD_ARGUMENT);
    }
    else {
        chunks = ((int64_t)inLen + chunk - 1) / chunk;
        const int64_t rest = (int64_t)inLen % chunk;
        const int64_t bound = ((int64_t)inLen / chunk) * (8 + block_bound(SNAPPY, chunk)) + (rest > 0 ? 8 + block_bound(SNAPPY, rest) : 0);
        if (bound > 0x7FFFFFFF) {
            st = mk_status(ACHIP_CLASS_INVALID_ARGUMENT, ACHIP_D_BAD_ARGUMENT);
        }
        else if ((int64_t)a.dstCap[stream] < bound) {
            st = mk_status(ACHIP_CLASS_OUTPUT_TOO_SMALL, ACHIP_D_HDP_MAX_OUTPUT);
        }
    }
    const int32_t n = st == 0 ? (int32_t)chunks : 0;
    const int32_t first = n > 0 ? atomicAdd(L.counters, n) : 0;
    if (st == 0 && (int64_t)first + n > MAX_CHUNKS) {  // (more than a million chunks in one call)
        st = mk_status(ACHIP_CLASS_INVALID_ARGUMENT, ACHIP_D_UNSUPPORTED);
    }
    L.sFirst[stream] = first;
    L.sCount[stream] = st == 0 ? n : 0;
    L.sStatus[stream] = st;
    for (int64_t k = 0; k < n && first + k < MAX_CHUNKS; k++) {
        L.bStream[first + k] = st == 0 ? stream : -1;
        L.bIndex[first + k] = (int32_t)k;
    }
}

__global__ void hadoop_seal_blocks_kernel(BlockList L)
{
    const int32_t allocated = L.counters[0];
    L.counters[1] = allocated < MAX_CHUNKS ? allocated : MAX_CHUNKS;
}

// LZ4: a wavefront per workgroup around its table in LDS.  Snappy (round 3): the two tiers of the block encoder (snappy_compress.hip) -- a 32 KB
// table allows five workgroups per CU, so a workgroup is four independent persistent wavefronts: wavefront 0 with the table in LDS, the others with a
// table slab each in memory, all drawing chunks from the one counter.
template <bool SNAPPY>
__global__ __launch_bounds__(SNAPPY ? 256 : 64) void hadoop_encode_kernel(BatchArgs a, BlockList L, int32_t bufferSize, uint16_t* slabs)
{
    // the codec's hash table: Snappy 16384 x u16; LZ4 4096 entries, u16 for chunks <= 64 KiB, i32 beyond
    __shared__ __attribute__((aligned(16))) uint8_t tableBytes[SNAPPY ? snc::MAX_HASH_TABLE_SIZE * 2 : lz4c::MAX_TABLE_SIZE * 4];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    uint16_t* const snappyTable = !SNAPPY || wave == 0 ? (uint16_t*)tableBytes : slabs + ((size_t)blockIdx.x * 3 + (wave - 1)) * snc::MAX_HASH_TABLE_SIZE;
    const int32_t total = L.counters[1];
    const int32_t chunk = input_max_size(SNAPPY, bufferSize);
    const int64_t worst = 8 + block_bound(SNAPPY, chunk);
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
        const int64_t pos = (int64_t)k * chunk;
        const int32_t inLen = a.srcLen[stream];
        const int32_t length = (int32_t)(inLen - pos < chunk ? inLen - pos : chunk);
        const uint8_t* block = a.srcBase + a.srcOff[stream] + pos;
        uint8_t* out = a.dstBase + a.dstOff[stream] + (int64_t)k * worst + 8;
        const int32_t cap = (int32_t)block_bound(SNAPPY, length);
        int32_t cst = 0, compressed = 0;
        if (SNAPPY) {
            if (wave == 0) {  // (two calls: the table's address space is part of the code)
                snappy_compress_buffer_mw((uint16_t*)tableBytes, block, length, out, cap, lane, cst, compressed);
            }
            else {
                snappy_compress_buffer_mw(snappyTable, block, length, out, cap, lane, cst, compressed);
            }
        }
        else if (length <= 65536) {
            compressed = lz4_compress_block_mw<uint16_t>(block, length, out, cap, (uint16_t*)tableBytes, lane, cst);
        }
        else {
            compressed = lz4_compress_block_mw<int32_t>(block, length, out, cap, (int32_t*)tableBytes, lane, cst);
        }
        wave_mem_order();
        if (lane == 0) {
            L.bSize[b] = cst == 0 ? compressed : cst;
        }
    }
}

template <bool SNAPPY>
__global__ __launch_bounds__(64) void hadoop_compact_kernel(BatchArgs a, BlockList L, int32_t bufferSize)
{
    const int lane = threadIdx.x;
    const int32_t chunk = input_max_size(SNAPPY, bufferSize);
    const int64_t worst = 8 + block_bound(SNAPPY, chunk);
    const int32_t stream = blockIdx.x;  // a wavefront per stream
    int32_t st = L.sStatus[stream];
    int32_t o = 0;
    if (st == 0) {
        uint8_t* out = a.dstBase + a.dstOff[stream];
        const int32_t inLen = a.srcLen[stream];
        const int32_t first = L.sFirst[stream], n = L.sCount[stream];
        for (int32_t k = 0; k < n; k++) {
            const int32_t size = L.bSize[first + k];
            if (size < 0) {
                st = size;  // (the block encoder's status)
                break;
            }
            const int64_t from = (int64_t)k * worst + 8;
            const int64_t pos = (int64_t)k * chunk;
            const int32_t length = (int32_t)(inLen - pos < chunk ? inLen - pos : chunk);
            wave_mem_order();
            if (from != (int64_t)o + 8) {
                group_copy<64>(out + o + 8, out + from, size, lane);  // to the left; every lane loads before it stores
                wave_mem_order();
            }
            if (lane == 0) {  // writeNextChunk :107-118 (written after the move: the fields of chunk k may lie inside chunk k - 1's worst-case slot)
                wr_be(out + o, length);
                wr_be(out + o + 4, size);
            }
            wave_mem_order();
            o += 8 + size;
        }
    }
    if (lane == 0) {
        a.outLen[stream] = st == 0 ? o : 0;
        a.status[stream] = st;
        a.errOffset[stream] = 0;
    }
}

}  // namespace hdp

hipError_t launch_lz4_decompress_rings(const BatchArgs& a, hipStream_t stream, int groupSize, int ringClass, const int32_t* mixedGroups);
int lz4_ring_group_for(int32_t nBlocks);
int snappy_ring_group_for(int32_t nBlocks);
hipError_t launch_lz4_sequence_sample(const BatchArgs& a, hipStream_t stream, int32_t* stats, int32_t minBlocks, int32_t shortLimit);
hipError_t launch_snappy_decompress_rings(const BatchArgs& a, hipStream_t stream, int groupSize, int ringClass, const int32_t* mixedGroups);
hipError_t launch_snappy_element_sample(const BatchArgs& a, hipStream_t stream, int32_t* stats, int32_t minBlocks, int32_t shortLimit);
hipError_t launch_lz4_mixed_groups(const BatchArgs& a, hipStream_t stream, int32_t* mixedGroups, int32_t minBlocks);
hipError_t launch_lz4_decompress_twopass(const BatchArgs& a, hipStream_t stream, void* scratch, int64_t scratchBytes, int groupSize, int ringClass, int execVariant, const int32_t* stats);
hipError_t launch_snappy_decompress_twopass(const BatchArgs& a, hipStream_t stream, void* scratch, int64_t scratchBytes, int groupSize, int ringClass, int execVariant, const int32_t* stats);
int64_t twopass_scratch_bytes(int32_t nBlocks, int64_t perBlock);

namespace {
constexpr int HDP_SERIAL_WAVES = 1024;
int64_t hdp_internal_bytes(int32_t bufferSize)
{
    const int64_t b = bufferSize > 262144 ? bufferSize : 262144;
    return (b + 8 + 63) & ~(int64_t)63;
}
}  // namespace

int64_t hadoop_decompress_scratch_bytes(int32_t nStreams, int32_t bufferSize)
{
    const int64_t n = nStreams < 1 ? 1 : nStreams;
    const int64_t waves = n < HDP_SERIAL_WAVES ? n : HDP_SERIAL_WAVES;
    return 4096 + n * 16 + 64 + (int64_t)hdp::MAX_CHUNKS * (8 * 3 + 4 * 5) + 4096 + waves * hdp_internal_bytes(bufferSize);
}

// variant 1 (default): the chunks through the ring decoders (with the probes' other choices behind them); variant 2 (round 2, written
// without a GPU at hand: not the default until measured): the chunks through the TWO-PASS decoders -- their record arena is
// sized by the chunk count, which only the device knows, so the host reads it back (one synchronisation) and asks `aux` for the arena;
// chunks whose records do not fit, and every chunk when there is no arena, take the ring decoder as in variant 1.
hipError_t launch_hadoop_decompress(const BatchArgs& a, hipStream_t stream, void* scratch, bool snappy, int32_t bufferSize, int variant, const AuxScratch* aux)
{
    if (a.nBlocks <= 0) {
        return hipSuccess;
    }
    uint8_t* base = (uint8_t*)scratch;
    int32_t* counters = (int32_t*)base;
    hipError_t e = hipMemsetAsync(counters, 0, 4096, stream);
    if (e != hipSuccess) return e;
    hdp::ChunkList L;
    uint8_t* p = base + 4096;
    const int64_t n = a.nBlocks;
    auto take = [&](int64_t bytes) {
        uint8_t* r = p;
        p += (bytes + 15) & ~(int64_t)15;
        return r;
    };
    L.counters = counters;
    L.sFirst = (int32_t*)take(4 * n);
    L.sCount = (int32_t*)take(4 * n);
    L.sOut = (int32_t*)take(4 * n);
    L.sSerial = (int32_t*)take(4 * n);
    const int64_t C = hdp::MAX_CHUNKS;
    L.cSrcOff = (int64_t*)take(8 * C);
    L.cDstOff = (int64_t*)take(8 * C);
    L.cErrOff = (int64_t*)take(8 * C);
    L.cSrcLen = (int32_t*)take(4 * C);
    L.cDstCap = (int32_t*)take(4 * C);
    L.cOutLen = (int32_t*)take(4 * C);
    L.cStatus = (int32_t*)take(4 * C);
    L.cExpect = (int32_t*)take(4 * C);
    uint8_t* internals = take(64);
    const int64_t internalMax = hdp_internal_bytes(bufferSize);
    const unsigned serialGrid = (unsigned)(a.nBlocks < HDP_SERIAL_WAVES ? a.nBlocks : HDP_SERIAL_WAVES);
    if (variant == 0) {  // one wavefront per stream only
        if (snappy) hipLaunchKernelGGL(hdp::hadoop_serial_decompress_kernel<true>, dim3(serialGrid), dim3(64), 0, stream, a, internals, (int32_t)internalMax, bufferSize, counters + 32, (const int32_t*)nullptr);
        else hipLaunchKernelGGL(hdp::hadoop_serial_decompress_kernel<false>, dim3(serialGrid), dim3(64), 0, stream, a, internals, (int32_t)internalMax, bufferSize, counters + 32, (const int32_t*)nullptr);
        return hipGetLastError();
    }
    const unsigned perStream = (unsigned)((a.nBlocks + 63) / 64);
    if (snappy) hipLaunchKernelGGL(hdp::hadoop_walk_kernel<true>, dim3(perStream), dim3(64), 0, stream, a, L);
    else hipLaunchKernelGGL(hdp::hadoop_walk_kernel<false>, dim3(perStream), dim3(64), 0, stream, a, L);
    hipLaunchKernelGGL(hdp::hadoop_seal_kernel, dim3(1), dim3(1), 0, stream, L);
    // the chunks as a batch of blocks whose size is known on the device only: launches are sized for the arrays
    BatchArgs c = a;
    c.srcOff = L.cSrcOff;
    c.srcLen = L.cSrcLen;
    c.dstOff = L.cDstOff;
    c.dstCap = L.cDstCap;
    c.outLen = L.cOutLen;
    c.status = L.cStatus;
    c.errOffset = L.cErrOff;
    c.nBlocks = hdp::MAX_CHUNKS;
    c.nBlocksDev = counters + 1;
    c.only = nullptr;
    c.onlyStats = nullptr;
    int32_t* stats = counters + 16;
    bool viaTwoPass = false;
    int32_t nChunksHost = -1;  // the chunk count once the host has read it (variants 2 and 3)
    if ((variant == 2 || variant == 3) && aux != nullptr && aux->get != nullptr) {
        // variant 3 (the default since round 3): the probes of the batched block API's auto mode (lz4_pick: mixed 16-chunk groups,
        // bytes per sampled sequence / element) run on the chunk list BEFORE the one synchronisation that reads the chunk count back, and
        // their verdict comes back with it -- text-like chunks (short sequences) go through the two-pass decoders, long copies
        // through the rings: measured (profiles/r03_notes.md, 1024 streams x 4 MiB) LZ4 818 / 83 GiB/s on fragments / corpus with the
        // rings, 271 / 158 with the two-pass decoders; Snappy 474 / 39 against 230 / 107
        int32_t head[20] = {0};
        if (variant == 3) {
            e = hipMemsetAsync(stats, 0, 4 * sizeof(int32_t), stream);
            if (e == hipSuccess) e = snappy ? launch_snappy_element_sample(c, stream, stats, 0, 0) : launch_lz4_sequence_sample(c, stream, stats, 0, 0);
            if (e != hipSuccess) return e;
        }
        e = hipMemcpyAsync(head, counters, sizeof(head), hipMemcpyDeviceToHost, stream);
        if (e != hipSuccess) return e;
        e = hipStreamSynchronize(stream);
        if (e != hipSuccess) return e;
        const int32_t nChunks = head[1];
        nChunksHost = nChunks;
        bool wantTwoPass = true;
        if (variant == 3) {
            // (only the sampled sequence lengths count here: a stream's last chunk is a short one, so "compressed sizes within a 16-chunk
            // group differ by 2x" -- the block API's sign of a mixed batch -- holds for every group of a batch of streams)
            const int32_t* v = head + 16;
            wantTwoPass = v[1] > 0 && (int64_t)v[2] < (int64_t)(snappy ? 6 : 12) * (int64_t)v[1];
        }
        if (!wantTwoPass) {
        }
        else if (nChunks > 0) {
            // records per chunk as for 64 KiB blocks (lz4_decompress_v7.hip twopass_scratch_bytes), scaled to the streams' chunk size
            const int64_t per64k = snappy ? 131072 : 98304;
            const int64_t perChunk = per64k * (((int64_t)(bufferSize > 65536 ? bufferSize : 65536) + 65535) / 65536);
            const int64_t bytes = twopass_scratch_bytes(nChunks, perChunk);
            void* arena = aux->get(aux->user, bytes);
            if (arena != nullptr) {
                BatchArgs t = c;
                t.nBlocks = nChunks;
                t.nBlocksDev = nullptr;
                e = snappy ? launch_snappy_decompress_twopass(t, stream, arena, bytes, 4, 0, 2, nullptr) : launch_lz4_decompress_twopass(t, stream, arena, bytes, 16, 0, 2, nullptr);
                if (e != hipSuccess) return e;
                viaTwoPass = true;
            }
        }
        else {
            viaTwoPass = true;  // (nothing listed)
        }
    }
    if (!viaTwoPass && nChunksHost >= 0) {
        // the host knows the chunk count (variant 3 chose the rings): one launch of the size that fits, no probes
        BatchArgs t = c;
        t.nBlocks = nChunksHost;
        t.nBlocksDev = nullptr;
        if (nChunksHost > 0) {
            e = snappy ? launch_snappy_decompress_rings(t, stream, snappy_ring_group_for(nChunksHost), 0, nullptr) : launch_lz4_decompress_rings(t, stream, lz4_ring_group_for(nChunksHost), 0, nullptr);
        }
    }
    else if (!viaTwoPass) {
    // LZ4: the ring decoder at two lane-group sizes: 4 lanes per chunk from 32768 chunks on, 16 below (721 -> 814 GiB/s fragments, 56 -> 83 corpus at 16384 chunks) (a stream's chunks are up to 256 KiB: a few
    // thousand of them at 4 lanes each leave most of the chip idle); the chunk count, known on the device only, picks one
    BatchArgs big = c, small = c;
    big.countLo = 32768;
    small.countHi = 32768;
    if (snappy) {
        e = launch_snappy_decompress_rings(c, stream, 4, 0, nullptr);  // (16 lanes per chunk measured slower for Snappy: 390 against 481 GiB/s)
    }
    else {
        e = launch_lz4_decompress_rings(big, stream, 4, 0, nullptr);
        if (e == hipSuccess) e = launch_lz4_decompress_rings(small, stream, 16, 0, nullptr);
    }
    }
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(hdp::hadoop_fold_kernel, dim3(perStream), dim3(64), 0, stream, a, L);
    // streams of any other shape, and those a chunk of which did not decode as declared
    if (snappy) hipLaunchKernelGGL(hdp::hadoop_serial_decompress_kernel<true>, dim3(serialGrid), dim3(64), 0, stream, a, internals, (int32_t)internalMax, bufferSize, counters + 32, (const int32_t*)L.sSerial);
    else hipLaunchKernelGGL(hdp::hadoop_serial_decompress_kernel<false>, dim3(serialGrid), dim3(64), 0, stream, a, internals, (int32_t)internalMax, bufferSize, counters + 32, (const int32_t*)L.sSerial);
    return hipGetLastError();
}

namespace {
constexpr int HADOOP_SNAPPY_WORKGROUPS = 256 * 5;  // five 32 KB LDS tables per CU, four wavefronts around each
constexpr int64_t HADOOP_SNAPPY_SLAB_BYTES = (int64_t)HADOOP_SNAPPY_WORKGROUPS * 3 * snc::MAX_HASH_TABLE_SIZE * 2 + 64;
}

int64_t hadoop_compress_scratch_bytes(int32_t nStreams)
{
    const int64_t n = nStreams < 1 ? 1 : nStreams;
    return 4096 + n * 12 + 64 + (int64_t)hdp::MAX_CHUNKS * 12 + 4096 + HADOOP_SNAPPY_SLAB_BYTES;
}

hipError_t launch_hadoop_compress(const BatchArgs& a, hipStream_t stream, void* scratch, bool snappy, int32_t bufferSize)
{
    if (a.nBlocks <= 0) {
        return hipSuccess;
    }
    uint8_t* base = (uint8_t*)scratch;
    int32_t* counters = (int32_t*)base;
    hipError_t e = hipMemsetAsync(counters, 0, 4096, stream);
    if (e != hipSuccess) return e;
    uint8_t* p = base + 4096;
    auto take = [&](int64_t bytes) {
        uint8_t* r = p;
        p += (bytes + 15) & ~(int64_t)15;
        return r;
    };
    hdp::BlockList L;
    const int64_t n = a.nBlocks;
    L.counters = counters;
    L.sFirst = (int32_t*)take(4 * n);
    L.sCount = (int32_t*)take(4 * n);
    L.sStatus = (int32_t*)take(4 * n);
    L.bStream = (int32_t*)take(4 * (int64_t)hdp::MAX_CHUNKS);
    L.bIndex = (int32_t*)take(4 * (int64_t)hdp::MAX_CHUNKS);
    L.bSize = (int32_t*)take(4 * (int64_t)hdp::MAX_CHUNKS);
    uint16_t* const slabs = (uint16_t*)take(HADOOP_SNAPPY_SLAB_BYTES - 64);
    const unsigned perStream = (unsigned)((a.nBlocks + 63) / 64);
    const unsigned encodeGrid = snappy ? HADOOP_SNAPPY_WORKGROUPS : 256 * 10;  // (LZ4: chunks beyond 64 KiB take the 16 KB table: ten wavefronts per CU)
    const unsigned compactGrid = (unsigned)a.nBlocks;
    if (snappy) {
        hipLaunchKernelGGL(hdp::hadoop_plan_kernel<true>, dim3(perStream), dim3(64), 0, stream, a, L, bufferSize);
        hipLaunchKernelGGL(hdp::hadoop_seal_blocks_kernel, dim3(1), dim3(1), 0, stream, L);
        hipLaunchKernelGGL(hdp::hadoop_encode_kernel<true>, dim3(encodeGrid), dim3(256), 0, stream, a, L, bufferSize, slabs);
        hipLaunchKernelGGL(hdp::hadoop_compact_kernel<true>, dim3(compactGrid), dim3(64), 0, stream, a, L, bufferSize);
    }
    else {
        hipLaunchKernelGGL(hdp::hadoop_plan_kernel<false>, dim3(perStream), dim3(64), 0, stream, a, L, bufferSize);
        hipLaunchKernelGGL(hdp::hadoop_seal_blocks_kernel, dim3(1), dim3(1), 0, stream, L);
        hipLaunchKernelGGL(hdp::hadoop_encode_kernel<false>, dim3(encodeGrid), dim3(64), 0, stream, a, L, bufferSize, slabs);
        hipLaunchKernelGGL(hdp::hadoop_compact_kernel<false>, dim3(compactGrid), dim3(64), 0, stream, a, L, bufferSize);
    }
    return hipGetLastError();
}

}  // namespace achip
