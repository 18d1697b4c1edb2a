// achip_abi.cpp -- host side of libaircompressor_hip.so: the C ABI of include/aircompressor_hip.h.
//
// Mirrors what the reference's FFM layer expects from a native codec library
// (M/internal/NativeLoader.java:66-117; M/lz4/Lz4Native.java:30-40): plain C symbols,
// int/long/pointer arguments, integer results.  Depends only on libamdhip64.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <thread>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "achip_device.h"

namespace achip {
hipError_t launch_lz4_decompress_rings(const BatchArgs& a, hipStream_t stream, int groupSize, int ringClass, const int32_t* mixedGroups);
int lz4_ring_group_for(int32_t nBlocks);
int snappy_ring_group_for(int32_t nBlocks);
hipError_t launch_lz4_decompress_twopass(const BatchArgs& a, hipStream_t stream, void* scratch, int64_t scratchBytes, int groupSize, int ringClass, int execVariant, const int32_t* stats);
int64_t twopass_scratch_bytes(int32_t nBlocks, int64_t perBlock);
// record arena per block of the two-pass decoders (8 bytes per record; lz4_decompress_v7.hip: text-like 64 KiB blocks make 6 000 .. 8 500 LZ4
// records, 8 500 .. 11 500 Snappy records), and the least it is worth running them with (blocks that do not fit go to the ring decoder)
constexpr int64_t LZ4_RECORD_BYTES_PER_BLOCK = 98304, LZ4_RECORD_BYTES_PER_BLOCK_MIN = 32768;
constexpr int64_t SNAPPY_RECORD_BYTES_PER_BLOCK = 131072, SNAPPY_RECORD_BYTES_PER_BLOCK_MIN = 49152;
hipError_t launch_snappy_decompress_twopass(const BatchArgs& a, hipStream_t stream, void* scratch, int64_t scratchBytes, int groupSize, int ringClass, int execVariant, const int32_t* stats);
hipError_t launch_lz4_mixed_groups(const BatchArgs& a, hipStream_t stream, int32_t* mixedGroups, int32_t minBlocks);
hipError_t launch_lz4_sequence_sample(const BatchArgs& a, hipStream_t stream, int32_t* stats, int32_t minBlocks, int32_t shortLimit);
hipError_t launch_snappy_decompress_rings(const BatchArgs& a, hipStream_t stream, int groupSize, int ringClass, const int32_t* mixedGroups);
hipError_t launch_snappy_element_sample(const BatchArgs& a, hipStream_t stream, int32_t* stats, int32_t minBlocks, int32_t shortLimit);
hipError_t launch_lz4_compress(const BatchArgs& a, hipStream_t stream, int variant, int maxSrcLenHint, void* scratch);
int64_t lz4_compress_scratch_bytes();
extern int g_lz4_mem_waves;
extern int g_lz4_tier_min_blocks;
hipError_t launch_snappy_compress(const BatchArgs& a, hipStream_t stream, int variant, void* scratch, bool fan);
hipError_t launch_blit(void* dst, const void* src, int64_t bytes, int workgroups, hipStream_t stream);
int64_t snappy_compress_scratch_bytes(int32_t nBlocks);
hipError_t launch_zstd_decompress(const BatchArgs& a, hipStream_t stream, void* scratch, int64_t scratchBytes, int variant, int32_t tileMax, const ZstdMbProvider* mbp);
hipError_t launch_zstd_compress(const BatchArgs& a, hipStream_t stream, void* scratch, int64_t scratchBytes, int variant);
hipError_t launch_zstd_stream_compress(const BatchArgs& a, hipStream_t stream, void* scratch, int chunked);
int64_t zstd_decompress_scratch_bytes(int32_t nBlocks, int32_t tileMax);
int64_t zstd_compress_scratch_bytes(int32_t nBlocks);
hipError_t launch_snappyframed_decompress(const BatchArgs& a, hipStream_t stream, void* scratch, int variant, const AuxScratch* aux);
int64_t snappyframed_decompress_scratch_bytes(int32_t nStreams);
hipError_t launch_snappyframed_compress(const BatchArgs& a, hipStream_t stream, void* scratch, int variant);
int64_t snappyframed_compress_scratch_bytes(int32_t nStreams);
hipError_t launch_hadoop_decompress(const BatchArgs& a, hipStream_t stream, void* scratch, bool snappy, int32_t bufferSize, int variant, const AuxScratch* aux);
int64_t hadoop_decompress_scratch_bytes(int32_t nStreams, int32_t bufferSize);
hipError_t launch_hadoop_compress(const BatchArgs& a, hipStream_t stream, void* scratch, bool snappy, int32_t bufferSize);
int64_t hadoop_compress_scratch_bytes(int32_t nStreams);
extern int g_zstd_pipe_exec;
extern int g_zstd_seq_waves;
extern int g_zstd_lit_items;
extern int g_snappy_mem_waves;
int64_t zstd_ostream_state_bytes();
int64_t zstd_ostream_slab_bytes();
hipError_t launch_zstd_ostream_step(hipStream_t stream, void* state, void* slab, const uint8_t* buf, int32_t offset, int32_t chunk, int32_t closing, uint8_t* out, int32_t outCap);
int64_t zstd_stream_carry_bytes();
void zstd_stream_carry_init(void* hostCarry);
int64_t zstd_stream_step_scratch_bytes(int32_t blocks);
hipError_t launch_zstd_stream_step(hipStream_t stream, void* scratch, int64_t scratchBytes, void* carryDev, const uint8_t* dSrc, int32_t srcLen, int32_t blocks, uint8_t* dOut,
                                   int32_t startPos, int32_t outLimit, int32_t closing, int32_t hasChecksum, uint32_t expected, int32_t* result);
extern int g_lz4_parse_mode;
extern int g_snappy_parse_mode;
hipError_t launch_lz4frame_decompress(const BatchArgs& a, hipStream_t stream, void* scratch, int variant, const AuxScratch* aux);
int64_t lz4frame_decompress_scratch_bytes(int32_t nItems, int variant);
hipError_t launch_lz4frame_compress(const BatchArgs& a, hipStream_t stream, void* scratch, int64_t scratchBytes);
int64_t lz4frame_compress_scratch_bytes(int32_t items, bool least);
hipError_t launch_mix_gather(const int32_t* perm, int32_t n, const BatchArgs& a, int64_t* gSrcOff, int32_t* gSrcLen, int64_t* gDstOff, int32_t* gDstCap, hipStream_t stream);
hipError_t launch_mix_scatter(const int32_t* perm, int32_t n, const int32_t* gOutLen, const int32_t* gStatus, const int64_t* gErr, const BatchArgs& a, hipStream_t stream);
hipError_t launch_xxh64_batch(const void* srcBase, const int64_t* srcOff, const int32_t* srcLen, int32_t n, uint64_t seed, int64_t* out, hipStream_t stream);
hipError_t launch_xxh32_batch(const void* srcBase, const int64_t* srcOff, const int32_t* srcLen, int32_t n, uint32_t seed, int32_t* out, hipStream_t stream);
}  // namespace achip

// What achip_ctx_set_option sets (and the bookkeeping of the last launch): a context's settings as a value, so that the helper contexts a mixed
// batch runs its buckets on (mix_lane) can take them over whole.
struct achip_options {
    int device = 0;
    int lz4dGroup = 0;       // ring decoder, lanes per block: 0 = by the batch size (4 from 32 768 blocks on -- the headline's form --, 16 below, 64 up to 4 096: lz4_ring_group_for), else 1 .. 64
    int snappydGroup = 0;    // likewise (64 up to 2 048 blocks, 16 below 16 384, 4 above: snappy_ring_group_for)
    int lz4dAutoMinBlocks = 4096;  // auto mode probes batches from this size on (smaller ones always take the rings)
    int lz4dVariant = 5;     // 5 = chosen on the device per batch (default: DESIGN 4c), 1 = LDS rings, a lane group per block (lz4_decompress_v2.hip), 7 = two passes: parse to records + a wavefront per block (lz4_decompress_v7.hip).  (4 / 6, a lane per block, lost to 7 on every batch they were built for -- 300 .. 330 GiB/s against 515 on corpus -- and were removed in round 4.)
    int snappydVariant = 5;  // 5 auto, 1 rings (snappy_decompress_v2.hip), 7 two passes (snappy_decompress_v5.hip), as for LZ4
    int smallBatchHint = 0;      // set by the host-pointer path for ONE launch: what a look at the first block's tokens says -- 1 short sequences, 2 long ones (0: nobody looked)
    int latencyMaxBlocks = 256;  // batches of at most this many blocks (a single block!) take the ring decoders' latency class: a wavefront and 128 KiB of LDS history per block
    int ringClass = 0;       // 0 = compact rings, 1 = large rings
    int lz4cVariant = 4;     // 4 = many matches per window of 64 positions (lz4_compress_mw.h; default since round 3: 25.8 against 18.2 GiB/s on corpus, 100 against 111 on fragments), 0 = serial probes, 1 = 64 probes per step (batch).  (3, the batch over an LDS input window, measured 17.2 against 18.2 GiB/s on corpus in round 3: removed)
    int snappycVariant = 4;  // THE DEFAULT IS 4 = two tiers, many matches per window (snappy_compress_mw.h; since round 3: 22.0 against 8.3 GiB/s on corpus, 65 against 74 on fragments); tested non-default variants: 0 = serial probes, 1 = 64 probes per step (batch), 2 = batch in two tiers: tables in LDS and in global memory.  (3, variant 2 over an LDS input window, measured 8.3 against 7.6 GiB/s for 2 and a third of variant 4: removed in round 4.)
    int zstddVariant = 1;  // 1 = five-stage pipeline (+ one-kernel decoder for its fallback list), 0 = one-kernel decoder only
    int zstdcVariant = 3;  // match kernel in window form (zstd_dfast_mw.h) + entropy kernel
    int hadoopBufferSize = 262144;        // Hadoop block streams: the streams' buffer size (Lz4HadoopStreams.java:30; io.compression.codec.*.buffersize)
    int lz4FrameDecompressVariant = 2;    // 2 = chosen per call by a probe of the sequence lengths (default: 75 / 13.4 GiB/s on fragments / corpus frames); 0 = a wavefront per item (75 / 7.8); 1 = the frames' blocks as one batch through the two-pass block decoder (22 / 13.4)
    int hadoopDecompressVariant = 3;      // 3 = chunk list, the block decoder chosen per call by a probe of the sequence lengths (default); 1 = always the rings; 2 = always the two-pass decoders; 0 = one wavefront per stream (profiles/r03_notes.md)
    int snappyFramedCompressVariant = 1;  // framed writer: 1 = block list + two-tier block encoder + compaction (default), 0 = one wavefront per stream
    int snappyFramedVariant = 3;  // framed reader: 3 = chunk list, the block decoder chosen per call by a probe of the element lengths (default); 1 = always the rings; 2 = always the two-pass decoder; 0 = one wavefront per stream
    int zstdTile = 65536;    // items per pass of the Zstd decode pipeline (halved automatically when its scratch cannot be allocated)
    int zstdStreamChunked = 1;     // 1: the stream writer takes streams from 4 MiB on as well (chunks flushed before close(), window slides: zstd_stream.hip; byte-identical with
                                   // the test suite's CPU restatement under tools/hostemu, not yet run on a GPU); 0: it refuses them (INVALID_ARGUMENT / ACHIP_D_UNSUPPORTED)
    int zstdStreamBlocks = 65536;  // 128 KiB blocks a pass of the pipeline's multi-block stages has room for (0: multi-block frames take the one-kernel decoder); ~20 GB of scratch, allocated when a batch first holds such frames (halved as often as it takes when the device cannot give that)
    int ringPad = 80;        // 64 bytes of far-match staging + 16: consecutive blocks start on different LDS banks
    int scratchPoison = -1;
    int32_t lastZstddBlocks = 0;  // achip_ctx_get_stat
    int lastZstddVariant = 0;
    int32_t lastAutoBlocks = 0;  // ... and this many blocks
    bool lastAutoIsLz4 = false;
    bool lastTwopass = false;   // the last decode was a two-pass one: its arena header leads the scratch
    bool lastLz4dAuto = false;  // the last LZ4 decode ran in auto mode: its probe count leads the scratch
    // Auto mode remembers (round 6): a call's probe statistics come back to pinned memory behind its kernels, without a wait; while the batches that follow have its
    // shape (codec, block count, the same source and destination buffers) the decoder the LAST ARRIVED statistics chose is the only one launched -- the other
    // decoder's kernels, launched to return at once, were ~65 us of a 8.2 ms headline call.  The probes still run in every call and go home, so a context whose
    // data changes character under one shape runs the wrong (slower, never incorrect: either decoder decodes any batch to the reference's bytes) decoder for as
    // many calls as it takes the first new statistics to arrive: one, for a caller that waits for its results.  (A first version probed one call in sixteen
    // and ran the rest blind: bench.py's own extras -- fragments, then text, same shape, same buffers -- decoded text on the rings for a whole measurement, 148
    // against 377 GiB/s.)  decompress.auto_remember = 0: both decoders are launched in every call, as until round 5.
    int autoRemember = 1;
    int32_t* autoPinned = nullptr;     // 8 words: the probe statistics of the call in flight
    hipEvent_t autoEv = nullptr;
    bool autoInFlight = false;
    int autoPendingFam = 0;
    int32_t autoPendingBlocks = 0;
    const void* autoPendingSrc = nullptr;
    const void* autoPendingDst = nullptr;
    int autoChoice[2] = {-1, -1};      // per codec family (0 LZ4, 1 Snappy): -1 unknown, 0 rings, 3 two passes
    int32_t autoBlocks[2] = {0, 0};
    const void* autoSrc[2] = {nullptr, nullptr};
    const void* autoDst[2] = {nullptr, nullptr};
    int lastRemembered = -1;           // the last decode ran on a remembered choice: that choice (decompress.choice reports it)
    int maxSrcLenHint = 0;
    int snappyFan = 1;     // snappy.compress.fan: 1 = the sub-blocks of buffers beyond 64 KiB are work units of their own (default), 0 = a buffer is one wavefront's work
    int execVariant = 2;     // two-pass decoders: 2 = the executor of achip_seqexec2.h (the only one)
    int mixConcurrent = 1;   // mixed batches: 1 = the three codec families side by side, each on a stream (and scratch) of its own -- a bucket's tail is a few long
                             // serial chains on a few CUs (a 10 MB file as ONE block: 0.4 s of one wavefront) --, 0 = every bucket in turn on the context's stream
};

struct achip_ctx : achip_options {
    hipStream_t stream = nullptr;
    // scratch for the zstd pipeline (grown on demand)
    void* scratch = nullptr;
    int64_t scratchBytes = 0;
    void* zstdMbScratch = nullptr;
    int64_t zstdMbScratchBytes = 0;
    // mixed batches, codec families side by side: a helper context for Snappy's and for Zstd's buckets (made when a batch first needs it: a stream and scratch of
    // its own, this context's options; LZ4's run on this context), and the events that order them behind the gather and in front of the scatter.  (Three streams, not
    // one per bucket: ROCm maps a process's streams onto GPU_MAX_HW_QUEUES = 4 hardware queues, and two long chains on one queue run one after the other --
    // profiles/r05_notes.md: six helper streams 1.22 s, with 8 queues 0.75.)
    achip_ctx* mixLane[3] = {};
    hipEvent_t mixGathered = nullptr, mixLaneDone[3] = {};
    // mixed batches (achip_mixed_batch): item permutation (pinned host + device) and the bucketed descriptor / result arrays
    int32_t* mixHost = nullptr;
    uint8_t* mixDev = nullptr;
    int64_t mixItems = 0;
    hipEvent_t mixUploaded = nullptr;  // the last permutation upload: the pinned buffer may be rewritten once it has completed
    // host-pointer batches (achip_batch_host / achip_mixed_batch_host): up to four staging slots, chunks pipelined over three streams, gather and
    // scatter on copy pools of their own
    static constexpr int kHostSlots = 8;
    int hostLookMaxBlocks = 0;  // host.look_max_blocks: chunks of up to this many blocks have their first tokens looked at on the host (long sequences: the rings at 64 lanes)
    int hostCopyLowPriority = 1;  // host.copy_priority: 1 = the pipeline's copy streams at the lowest stream priority (to be set before the first host-pointer batch)
    int hostRamp = 1;  // host.ramp: 1 = smaller chunks at a batch's start and end (default), 0 = chunks of host.chunk_bytes throughout
    int hostSlots = 8;  // host.slots: staging slots the host-pointer pipeline uses (2 .. kHostSlots; round 6: 8 -- with 4 the gather thread waited for a slot 20 of a call's 34 ms)
    struct CopyPool* pool = nullptr;     // gather: the caller's inputs -> pinned slot
    struct CopyPool* poolOut = nullptr;  // scatter: pinned slot -> the caller's outputs
    uint8_t* slotHost[kHostSlots] = {};  // pinned
    uint8_t* slotDev[kHostSlots] = {};
    int64_t slotBytes = 0;
    int slotCount = 0;
    hipStream_t copyIn = nullptr, copyOut = nullptr;
    hipEvent_t evH2D[kHostSlots] = {}, evK[kHostSlots] = {}, evD2H[kHostSlots] = {};
    int hostBlit = 0;          // host.blit: bit 0 = the pipeline's uploads by a copy kernel instead of hipMemcpyAsync, bit 1 = its downloads
    int hostBlitGroups = 128;  // host.blit_groups: workgroups of that kernel
    int64_t hostChunkBytes = 192 << 20;   // staging bytes (inputs + output capacities) per pipeline chunk: ~2000 blocks of 64 KiB -- a chunk's kernels
                                         // take a block's serial chain (~1-2 ms) however few blocks it holds, so a chunk must be worth that long on the
                                         // link.  Round 6 (profiles/r06_hostsweep.txt, r06_host_timeline.txt): with eight slots, the copy streams at the lowest
                                         // priority and smaller chunks at both ends 192 MiB gives 40-42 GiB/s where round 5's 96 MiB over four slots gave 26-30
                                         // on the same box (the 48 MiB of rounds 1-4 over two slots: 8.7)
    int hostCopyThreads = 0;             // per copy pool; 0 = hardware threads / 16, 2 .. 8 (4 and 8 measured best; 32 no better: the scatter is
                                         // bound by the host's memory system, not by the thread count)
    // achip_ctx_get_stat("host.*"): where the last host-pointer batch of several chunks spent its wall time (microseconds)
    int64_t hostGatherUs = 0, hostScatterUs = 0, hostWaitSlotUs = 0, hostWaitDownloadUs = 0, hostChunks = 0, hostTotalUs = 0;
    // staging for the one-shot hashers (grown on demand)
    uint8_t* hostStage = nullptr;  // pinned
    uint8_t* devStage = nullptr;
    int64_t stageBytes = 0;
};

namespace {

void destroy_host_path(achip_ctx* ctx);

thread_local std::string g_lastError;

int32_t device_failure(const char* what, hipError_t e)
{
    char buf[256];
    snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
    g_lastError = buf;
    return ACHIP_STATUS(ACHIP_CLASS_DEVICE, ACHIP_D_HIP_ERROR);
}

#define HIP_TRY(expr)                             \
    do {                                          \
        hipError_t e_ = (expr);                   \
        if (e_ != hipSuccess) {                   \
            return device_failure(#expr, e_);     \
        }                                         \
    } while (0)

int32_t bad_argument(const char* what)
{
    g_lastError = what;
    return ACHIP_STATUS(ACHIP_CLASS_INVALID_ARGUMENT, ACHIP_D_BAD_ARGUMENT);
}

achip::BatchArgs make_args(const void* srcBase, const int64_t* srcOff, const int32_t* srcLen, void* dstBase, const int64_t* dstOff,
                           const int32_t* dstCap, int32_t* outLen, int32_t* status, int64_t* errOffset, int32_t nBlocks)
{
    achip::BatchArgs a;
    a.srcBase = (const uint8_t*)srcBase;
    a.srcOff = srcOff;
    a.srcLen = srcLen;
    a.dstBase = (uint8_t*)dstBase;
    a.dstOff = dstOff;
    a.dstCap = dstCap;
    a.outLen = outLen;
    a.status = status;
    a.errOffset = errOffset;
    a.nBlocks = nBlocks;
    a.ringPad = 0;
    a.nBlocksDev = nullptr;
    a.only = nullptr;
    a.onlyStats = nullptr;
    a.onlyShortLimit = 12;
    return a;
}

int32_t ensure_scratch(achip_ctx* ctx, int64_t bytes);
int32_t grow_scratch_keeping_old(achip_ctx* ctx, int64_t bytes);
// The two-pass decoders' scratch (lead bytes of probe statistics + header, meta and record arena): the full arena (`perBlock` bytes of
// records per block) if the device has it to spare -- never more than half of what is free right now beyond what the context already
// holds, so that one large batch does not take the device from its other users --, else what is there down to `perBlockMin` (blocks whose
// records do not fit are decoded by the ring decoder: the parse kernels hand them over per block).  Returns 1 with the scratch in
// place, 0 when not even the minimum could be had -- the caller then runs the ring decoder alone, which needs no scratch: a batch that
// decoded before the two-pass decoders existed still decodes on a busy device -- and < 0 for an error that is not about memory.
// Record bytes per block for the few-blocks route (a batch below the size auto mode probes from, its block sizes known on the device only): such blocks may be whole
// files -- a 4 MiB text block makes 6 MiB of records where the block codec's 64 KiB blocks make 96 KiB --, so the arena is sized as a whole: a GiB over however few
// blocks there are (as ever at most half of what the device has free; blocks that still do not fit go to the ring decoder, now at 64 lanes each).
int64_t few_blocks_record_bytes(int32_t nBlocks, int64_t perBlock)
{
    if (nBlocks >= 4096 || nBlocks <= 0) return perBlock;
    return std::max<int64_t>(perBlock, ((1LL << 30) / nBlocks) & ~4095LL);
}

// Auto mode's memory (achip_ctx::autoRemember).  auto_remembered: the decoder to launch alone for this batch (0 rings, 3 two passes), or -1: launch both.  First takes
// in what an earlier call sent home, if it has arrived (an event query, never a wait).
int auto_remembered(achip_ctx* ctx, int fam, const achip::BatchArgs& a)
{
    if (ctx->autoInFlight && hipEventQuery(ctx->autoEv) == hipSuccess) {
        const int32_t* v = ctx->autoPinned;
        const int pf = ctx->autoPendingFam;
        const bool mixed = (int64_t)v[0] * 4 > (ctx->autoPendingBlocks + 15) / 16;  // the rule of lz4_pick (achip_device.h)
        const bool pooledShort = v[1] > 0 && (int64_t)v[2] < (pf == 0 ? 12 : 6) * (int64_t)v[1];
        const bool isShort = v[5] > 0 ? (int64_t)v[4] * 3 > (int64_t)v[5] : pooledShort;
        ctx->autoChoice[pf] = (mixed || isShort) ? 3 : 0;
        ctx->autoBlocks[pf] = ctx->autoPendingBlocks;
        ctx->autoSrc[pf] = ctx->autoPendingSrc;
        ctx->autoDst[pf] = ctx->autoPendingDst;
        ctx->autoInFlight = false;
    }
    else if (ctx->autoInFlight) {
        (void)hipGetLastError();  // (hipErrorNotReady is not an error)
    }
    if (ctx->autoRemember == 0 || ctx->autoChoice[fam] < 0 || a.nBlocksDev != nullptr || a.only != nullptr) return -1;
    if (ctx->autoBlocks[fam] != a.nBlocks || ctx->autoSrc[fam] != (const void*)a.srcBase || ctx->autoDst[fam] != (const void*)a.dstBase) return -1;
    ctx->lastRemembered = ctx->autoChoice[fam];
    return ctx->autoChoice[fam];
}
// behind a call's kernels: its probe statistics on their way to pinned memory (nothing waits for them; one set in flight at a time)
void auto_send_home(achip_ctx* ctx, int fam, const achip::BatchArgs& a, const int32_t* stats)
{
    if (ctx->autoRemember == 0 || ctx->autoInFlight || a.nBlocksDev != nullptr || a.only != nullptr) return;
    if (!ctx->autoPinned) {
        if (hipHostMalloc((void**)&ctx->autoPinned, 64, hipHostMallocDefault) != hipSuccess || hipEventCreateWithFlags(&ctx->autoEv, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            if (ctx->autoPinned) (void)hipHostFree(ctx->autoPinned);
            ctx->autoPinned = nullptr;
            ctx->autoRemember = 0;  // (no memory for it: both decoders in every call)
            return;
        }
    }
    if (hipMemcpyAsync(ctx->autoPinned, stats, 32, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipEventRecord(ctx->autoEv, ctx->stream) != hipSuccess) {
        (void)hipGetLastError();
        return;
    }
    ctx->autoInFlight = true;
    ctx->autoPendingFam = fam;
    ctx->autoPendingBlocks = a.nBlocks;
    ctx->autoPendingSrc = a.srcBase;
    ctx->autoPendingDst = a.dstBase;
}

int32_t ensure_twopass_scratch(achip_ctx* ctx, int64_t lead, int32_t nBlocks, int64_t perBlock, int64_t perBlockMin)
{
    const int64_t want = lead + achip::twopass_scratch_bytes(nBlocks, perBlock);
    if (want <= ctx->scratchBytes) {
        return 1;
    }
    const int64_t atLeast = lead + achip::twopass_scratch_bytes(nBlocks, perBlockMin);
    HIP_TRY(hipSetDevice(ctx->device));
    size_t freeB = 0, totalB = 0;
    int64_t ask = want;
    if (hipMemGetInfo(&freeB, &totalB) == hipSuccess) {
        const int64_t room = ctx->scratchBytes + (int64_t)(freeB / 2);
        ask = std::min(want, std::max(room, atLeast));
    }
    else {
        (void)hipGetLastError();
    }
    if (ask <= ctx->scratchBytes) {
        return ctx->scratchBytes >= atLeast ? 1 : 0;
    }
    // Grow WITHOUT giving up what is there: the new buffer is allocated first and the old one freed only when that worked, so a
    // context never loses a working scratch to a failed request.  Only a context whose scratch is below the minimum anyway lets it go and
    // asks again (the device may have room for one of the two, not both).
    if (grow_scratch_keeping_old(ctx, ask) == 0) {
        return 1;
    }
    if (ask > atLeast && ctx->scratchBytes < atLeast && grow_scratch_keeping_old(ctx, atLeast) == 0) {
        return 1;
    }
    if (ctx->scratchBytes >= atLeast) {
        g_lastError.clear();
        return 1;  // what is there serves (fewer records per block: more blocks go to the ring decoder -- decompress.twopass_fallback_blocks)
    }
    if (ensure_scratch(ctx, atLeast) == 0) {  // (frees the old scratch first)
        return 1;
    }
    g_lastError.clear();
    return 0;
}

int32_t ensure_scratch(achip_ctx* ctx, int64_t bytes)
{
    if (bytes <= ctx->scratchBytes) {
        return 0;
    }
    HIP_TRY(hipSetDevice(ctx->device));
    if (ctx->scratch) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        HIP_TRY(hipFree(ctx->scratch));
        ctx->scratch = nullptr;
        ctx->scratchBytes = 0;
    }
    const hipError_t e = hipMalloc(&ctx->scratch, (size_t)bytes);
    if (e != hipSuccess) {
        ctx->scratch = nullptr;
        (void)hipGetLastError();  // not sticky: the caller may retry with a smaller request
        return device_failure("hipMalloc(scratch)", e);
    }
    ctx->scratchBytes = bytes;
    return 0;
}

// bytes > what is there: allocates the larger buffer, then -- only then -- frees the old one.  Non-zero (and nothing changed) when the device
// cannot hold both.
int32_t grow_scratch_keeping_old(achip_ctx* ctx, int64_t bytes)
{
    if (bytes <= ctx->scratchBytes) {
        return 0;
    }
    HIP_TRY(hipSetDevice(ctx->device));
    void* fresh = nullptr;
    const hipError_t e = hipMalloc(&fresh, (size_t)bytes);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return -1;
    }
    if (ctx->scratch) {
        const hipError_t s = hipStreamSynchronize(ctx->stream);
        if (s != hipSuccess) {
            (void)hipFree(fresh);
            return device_failure("hipStreamSynchronize", s);
        }
        (void)hipFree(ctx->scratch);
    }
    ctx->scratch = fresh;
    ctx->scratchBytes = bytes;
    return 0;
}

// scratch of the Zstd pipeline's multi-block stages (achip::ZstdMbProvider::get; the stream is idle when the stages ask)
void* zstd_mb_scratch(void* user, int64_t bytes)
{
    achip_ctx* ctx = (achip_ctx*)user;
    if (bytes <= ctx->zstdMbScratchBytes) {
        return ctx->zstdMbScratch;
    }
    if (hipSetDevice(ctx->device) != hipSuccess) {
        return nullptr;
    }
    if (ctx->zstdMbScratch) {
        (void)hipFree(ctx->zstdMbScratch);
        ctx->zstdMbScratch = nullptr;
        ctx->zstdMbScratchBytes = 0;
    }
    if (hipMalloc(&ctx->zstdMbScratch, (size_t)bytes) != hipSuccess) {
        ctx->zstdMbScratch = nullptr;
        (void)hipGetLastError();
        return nullptr;
    }
    ctx->zstdMbScratchBytes = bytes;
    return ctx->zstdMbScratch;
}

int32_t ensure_stage(achip_ctx* ctx, int64_t bytes)
{
    if (bytes <= ctx->stageBytes) {
        return 0;
    }
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (ctx->hostStage) {
        HIP_TRY(hipHostFree(ctx->hostStage));
        ctx->hostStage = nullptr;
    }
    if (ctx->devStage) {
        HIP_TRY(hipFree(ctx->devStage));
        ctx->devStage = nullptr;
    }
    ctx->stageBytes = 0;
    int64_t want = std::max<int64_t>(bytes, 1 << 20);
    HIP_TRY(hipHostMalloc((void**)&ctx->hostStage, (size_t)want, hipHostMallocDefault));
    HIP_TRY(hipMalloc((void**)&ctx->devStage, (size_t)want));
    ctx->stageBytes = want;
    return 0;
}

int32_t launch_op(int32_t op, achip_ctx* ctx, const achip::BatchArgs& args)
{
    if (!ctx) {
        return bad_argument("ctx is null");
    }
    achip::BatchArgs a = args;
    a.ringPad = ctx->ringPad;
    if (op == ACHIP_OP_ZSTD_COMPRESS) a.ringPad = ctx->zstdcVariant == 1 ? 1 : (ctx->zstdcVariant == 3 ? 3 : 0);  // the encoder variant rides in the spare field
    if (a.nBlocks < 0) {
        return bad_argument("nBlocks < 0");
    }
    if (a.nBlocks == 0) {
        return 0;
    }
    HIP_TRY(hipSetDevice(ctx->device));
    hipError_t e = hipSuccess;
    ctx->lastLz4dAuto = false;
    ctx->lastTwopass = false;
    ctx->lastRemembered = -1;
    switch (op) {
        case ACHIP_OP_LZ4_DECOMPRESS: {
            const int lz4Group = ctx->lz4dGroup > 0 ? ctx->lz4dGroup : (a.nBlocksDev != nullptr ? 4 : achip::lz4_ring_group_for(a.nBlocks));
            if (ctx->lz4dVariant == 5 && a.nBlocks >= ctx->lz4dAutoMinBlocks) {
                // auto: the choice is made on the device (no host round trip): probes count the mixed 16-block groups and sample the
                // sequence lengths, every candidate decoder is launched and the ones not chosen return at once.  Mixed or short-sequence
                // batches go to the two-pass decoder (parse to records + a wavefront per block), the rest to the rings.
                // scratch: [probe statistics: the first 4 KiB][two-pass header, meta, arena]
                const int32_t r = ensure_twopass_scratch(ctx, 4096, a.nBlocks, achip::LZ4_RECORD_BYTES_PER_BLOCK, achip::LZ4_RECORD_BYTES_PER_BLOCK_MIN);
                if (r < 0) return r;
                if (r == 0) {  // no room for records on this device right now: the rings alone
                    e = achip::launch_lz4_decompress_rings(a, ctx->stream, lz4Group, ctx->ringClass, nullptr);
                    break;
                }
                int32_t* stats = (int32_t*)ctx->scratch;
                ctx->lastZstddBlocks = 0;
                const int remembered = auto_remembered(ctx, 0, a);
                ctx->lastLz4dAuto = true;
                ctx->lastAutoBlocks = a.nBlocks;
                ctx->lastAutoIsLz4 = true;
                e = achip::launch_lz4_mixed_groups(a, ctx->stream, stats, 0);
                if (e == hipSuccess) e = hipMemsetAsync(stats + 3, 1, 1, ctx->stream);  // stats[3] = 1: the two-pass scheme (achip_device.h lz4_pick)
                if (e == hipSuccess) e = achip::launch_lz4_sequence_sample(a, ctx->stream, stats, 0, 12);
                // (remembered: that decoder alone and whatever this call's probes say -- they are for the calls to come)
                if (e == hipSuccess && remembered != 3) e = achip::launch_lz4_decompress_rings(a, ctx->stream, lz4Group, ctx->ringClass, remembered == 0 ? nullptr : stats);
                if (remembered != 0) {
                    ctx->lastTwopass = true;
                    if (e == hipSuccess) e = achip::launch_lz4_decompress_twopass(a, ctx->stream, (uint8_t*)ctx->scratch + 4096, ctx->scratchBytes - 4096, lz4Group, ctx->ringClass, ctx->execVariant, remembered == 3 ? nullptr : stats);
                }
                if (e == hipSuccess) auto_send_home(ctx, 0, a, stats);
                break;
            }
            // FEW blocks (at most decompress.latency_max_blocks; a single block is the literal Lz4HipDecompressor.decompress): nothing hides a lone block's chain,
            // so the choice is by what one 64 KiB block costs (profiles/r05_single_block_latency.txt): long sequences -- the ring decoders' latency class
            // (0.64 ms; the two passes 1.0); short ones or unknown -- the two passes with the wavefront-per-block parser (text 1.3 ms; the latency class 5.1)
            // ... and so does every batch below the size auto mode probes from (4 096): such a batch of LARGE blocks -- files as single blocks, a frame's 4 MiB blocks --
            // is the compact rings' worst case (four lanes per block, 83 % of a text block's matches a memory round trip each: the 263-item LZ4 bucket of the mixed corpus batch
            // took 526 ms, its two largest files alone through the two passes 80), and on long sequences the two passes cost about what the rings do at these sizes
            const bool fewBlocks = ctx->lz4dVariant == 5 && a.nBlocks < ctx->lz4dAutoMinBlocks && a.nBlocksDev == nullptr && a.only == nullptr;
            const int hint = ctx->smallBatchHint;
            ctx->smallBatchHint = 0;
            if (ctx->lz4dVariant == 7 || (fewBlocks && hint != 2)) {  // two passes: parse to records, a wavefront per block executes them (lz4_decompress_v7.hip)
                const int32_t r = ensure_twopass_scratch(ctx, 0, a.nBlocks, few_blocks_record_bytes(a.nBlocks, achip::LZ4_RECORD_BYTES_PER_BLOCK), achip::LZ4_RECORD_BYTES_PER_BLOCK_MIN);
                if (r < 0) return r;
                if (r == 0) {
                    e = achip::launch_lz4_decompress_rings(a, ctx->stream, lz4Group, ctx->ringClass, nullptr);
                    break;
                }
                ctx->lastTwopass = true;
                e = achip::launch_lz4_decompress_twopass(a, ctx->stream, ctx->scratch, ctx->scratchBytes, lz4Group, ctx->ringClass, ctx->execVariant, nullptr);
                break;
            }
            e = achip::launch_lz4_decompress_rings(a, ctx->stream, lz4Group, ctx->ringClass == 0 && a.nBlocks <= ctx->latencyMaxBlocks ? 3 : ctx->ringClass, nullptr);
            break;
        }
        case ACHIP_OP_LZ4_COMPRESS: {
            if (ctx->lz4cVariant == 4 && achip::g_lz4_mem_waves > 0) {  // (the two-tier kernel's table slabs)
                const int32_t r = ensure_scratch(ctx, achip::lz4_compress_scratch_bytes());
                if (r < 0) return r;
            }
            e = achip::launch_lz4_compress(a, ctx->stream, ctx->lz4cVariant, ctx->maxSrcLenHint, ctx->scratch);
            break;
        }
        case ACHIP_OP_SNAPPY_DECOMPRESS: {
            const int snappyGroup = ctx->snappydGroup > 0 ? ctx->snappydGroup : (a.nBlocksDev != nullptr ? 4 : achip::snappy_ring_group_for(a.nBlocks));
            if (ctx->snappydVariant == 5 && a.nBlocks >= ctx->lz4dAutoMinBlocks) {  // auto, as for LZ4
                const int32_t r = ensure_twopass_scratch(ctx, 4096, a.nBlocks, achip::SNAPPY_RECORD_BYTES_PER_BLOCK, achip::SNAPPY_RECORD_BYTES_PER_BLOCK_MIN);
                if (r < 0) return r;
                if (r == 0) {
                    e = achip::launch_snappy_decompress_rings(a, ctx->stream, snappyGroup, ctx->ringClass, nullptr);
                    break;
                }
                int32_t* stats = (int32_t*)ctx->scratch;
                ctx->lastZstddBlocks = 0;
                const int remembered = auto_remembered(ctx, 1, a);
                ctx->lastLz4dAuto = true;
                ctx->lastAutoBlocks = a.nBlocks;
                ctx->lastAutoIsLz4 = false;
                e = achip::launch_lz4_mixed_groups(a, ctx->stream, stats, 0);
                if (e == hipSuccess) e = hipMemsetAsync(stats + 3, 1, 1, ctx->stream);
                if (e == hipSuccess) e = achip::launch_snappy_element_sample(a, ctx->stream, stats, 0, 6);
                if (e == hipSuccess && remembered != 3) e = achip::launch_snappy_decompress_rings(a, ctx->stream, snappyGroup, ctx->ringClass, remembered == 0 ? nullptr : stats);
                if (remembered != 0) {
                    ctx->lastTwopass = true;
                    if (e == hipSuccess) e = achip::launch_snappy_decompress_twopass(a, ctx->stream, (uint8_t*)ctx->scratch + 4096, ctx->scratchBytes - 4096, snappyGroup, ctx->ringClass, ctx->execVariant, remembered == 3 ? nullptr : stats);
                }
                if (e == hipSuccess) auto_send_home(ctx, 1, a, stats);
                break;
            }
            // (few blocks, and every batch below the size auto mode probes from: as for LZ4 -- the two passes with the wavefront-per-block parser unless the host looked and
            // saw long elements, which take the latency class: one 64 KiB text block 7.2 ms with a lane parsing it, profiles/r05_single_block_latency.txt for what it is now)
            const bool fewSnappy = ctx->snappydVariant == 5 && a.nBlocks < ctx->lz4dAutoMinBlocks && a.nBlocksDev == nullptr && a.only == nullptr && ctx->smallBatchHint != 2;
            ctx->smallBatchHint = 0;
            if (ctx->snappydVariant == 7 || fewSnappy) {  // two passes (snappy_decompress_v5.hip)
                const int32_t r = ensure_twopass_scratch(ctx, 0, a.nBlocks, few_blocks_record_bytes(a.nBlocks, achip::SNAPPY_RECORD_BYTES_PER_BLOCK), achip::SNAPPY_RECORD_BYTES_PER_BLOCK_MIN);
                if (r < 0) return r;
                if (r == 0) {
                    e = achip::launch_snappy_decompress_rings(a, ctx->stream, snappyGroup, ctx->ringClass, nullptr);
                    break;
                }
                ctx->lastTwopass = true;
                e = achip::launch_snappy_decompress_twopass(a, ctx->stream, ctx->scratch, ctx->scratchBytes, snappyGroup, ctx->ringClass, ctx->execVariant, nullptr);
                break;
            }
            e = achip::launch_snappy_decompress_rings(a, ctx->stream, snappyGroup, ctx->ringClass == 0 && a.nBlocks <= ctx->latencyMaxBlocks ? 3 : ctx->ringClass, nullptr);
            break;
        }
        case ACHIP_OP_SNAPPY_COMPRESS: {
            if (ctx->snappycVariant >= 2) {
                int32_t r = ensure_scratch(ctx, achip::snappy_compress_scratch_bytes(a.nBlocks));
                if (r < 0) return r;
            }
            // (buffers beyond 64 KiB: their independent sub-blocks side by side -- unless the caller, or the host-pointer path that has seen the lengths, says there are none)
            e = achip::launch_snappy_compress(a, ctx->stream, ctx->snappycVariant, ctx->scratch, ctx->snappyFan != 0 && !(ctx->maxSrcLenHint > 0 && ctx->maxSrcLenHint <= 65536));
            break;
        }
        case ACHIP_OP_ZSTD_DECOMPRESS: {
            // pipeline scratch scales with the tile (items per pass, <= 65536: ~17 GB); when the device cannot give that much,
            // smaller tiles are tried before giving up (the one-kernel decoder's 270 MB are always part of it)
            int32_t r = ensure_scratch(ctx, achip::zstd_decompress_scratch_bytes(a.nBlocks, ctx->zstdTile));
            while (r < 0 && ctx->zstdTile > 1024 && ctx->zstdTile >= a.nBlocks / 64) {
                ctx->zstdTile /= 2;
                r = ensure_scratch(ctx, achip::zstd_decompress_scratch_bytes(a.nBlocks, ctx->zstdTile));
            }
            if (r < 0) return r;
            const achip::ZstdMbProvider mbp{zstd_mb_scratch, ctx, ctx->zstdStreamBlocks};
            e = achip::launch_zstd_decompress(a, ctx->stream, ctx->scratch, ctx->scratchBytes, ctx->zstddVariant, ctx->zstdTile, ctx->zstdStreamBlocks >= 16 ? &mbp : nullptr);
            ctx->lastZstddBlocks = a.nBlocks;
            ctx->lastZstddVariant = ctx->zstddVariant;
            break;
        }
        case ACHIP_OP_LZ4FRAME_DECOMPRESS: {
            int32_t r = ensure_scratch(ctx, achip::lz4frame_decompress_scratch_bytes(a.nBlocks, ctx->lz4FrameDecompressVariant));
            if (r < 0) return r;
            const achip::AuxScratch aux{zstd_mb_scratch, ctx};
            e = achip::launch_lz4frame_decompress(a, ctx->stream, ctx->scratch, ctx->lz4FrameDecompressVariant, &aux);
            break;
        }
        case ACHIP_OP_SNAPPYFRAMED_DECOMPRESS: {
            int32_t r = ensure_scratch(ctx, achip::snappyframed_decompress_scratch_bytes(a.nBlocks));
            if (r < 0) return r;
            const achip::AuxScratch aux{zstd_mb_scratch, ctx};
            e = achip::launch_snappyframed_decompress(a, ctx->stream, ctx->scratch, ctx->snappyFramedVariant, &aux);
            break;
        }
        case ACHIP_OP_SNAPPYFRAMED_COMPRESS: {
            int32_t r = ensure_scratch(ctx, achip::snappyframed_compress_scratch_bytes(a.nBlocks));
            if (r < 0) return r;
            e = achip::launch_snappyframed_compress(a, ctx->stream, ctx->scratch, ctx->snappyFramedCompressVariant);
            break;
        }
        case ACHIP_OP_LZ4HADOOP_DECOMPRESS:
        case ACHIP_OP_SNAPPYHADOOP_DECOMPRESS: {
            int32_t r = ensure_scratch(ctx, achip::hadoop_decompress_scratch_bytes(a.nBlocks, ctx->hadoopBufferSize));
            if (r < 0) return r;
            const achip::AuxScratch aux{zstd_mb_scratch, ctx};  // (the context's second, lazily grown buffer: shared with the Zstd multi-block stages)
            e = achip::launch_hadoop_decompress(a, ctx->stream, ctx->scratch, op == ACHIP_OP_SNAPPYHADOOP_DECOMPRESS, ctx->hadoopBufferSize, ctx->hadoopDecompressVariant, &aux);
            break;
        }
        case ACHIP_OP_LZ4HADOOP_COMPRESS:
        case ACHIP_OP_SNAPPYHADOOP_COMPRESS: {
            int32_t r = ensure_scratch(ctx, achip::hadoop_compress_scratch_bytes(a.nBlocks));
            if (r < 0) return r;
            e = achip::launch_hadoop_compress(a, ctx->stream, ctx->scratch, op == ACHIP_OP_SNAPPYHADOOP_COMPRESS, ctx->hadoopBufferSize);
            break;
        }
        case ACHIP_OP_LZ4FRAME_COMPRESS: {
            // a slab per resident wavefront; when the device cannot give the full set the launch runs with fewer wavefronts
            if (grow_scratch_keeping_old(ctx, achip::lz4frame_compress_scratch_bytes(a.nBlocks, false)) != 0) {
                g_lastError.clear();
                int32_t r = ensure_scratch(ctx, achip::lz4frame_compress_scratch_bytes(a.nBlocks, true));
                if (r < 0) return r;
            }
            e = achip::launch_lz4frame_compress(a, ctx->stream, ctx->scratch, ctx->scratchBytes);
            break;
        }
        case ACHIP_OP_ZSTDSTREAM_COMPRESS: {
            int32_t r = ensure_scratch(ctx, achip::zstd_compress_scratch_bytes(a.nBlocks));
            if (r < 0) return r;
            e = achip::launch_zstd_stream_compress(a, ctx->stream, ctx->scratch, ctx->zstdStreamChunked);
            break;
        }
        case ACHIP_OP_ZSTD_COMPRESS: {
            int32_t r = ensure_scratch(ctx, achip::zstd_compress_scratch_bytes(a.nBlocks));
            if (r < 0) return r;
            if (ctx->scratchPoison >= 0) {  // debugging aid: expose reads of uninitialised scratch
                HIP_TRY(hipMemsetAsync(ctx->scratch, ctx->scratchPoison, (size_t)ctx->scratchBytes, ctx->stream));
            }
            e = achip::launch_zstd_compress(a, ctx->stream, ctx->scratch, ctx->scratchBytes, ctx->zstdcVariant);
            break;
        }
        default: return bad_argument("unknown codecOp");
    }
    if (e != hipSuccess) {
        return device_failure("kernel launch", e);
    }
    return 0;
}

struct DetailText {
    int32_t detail;
    const char* text;
};
const DetailText kDetailText[] = {
    {ACHIP_D_GENERIC, "Unknown error"},
    {ACHIP_D_LZ4_INPUT_EMPTY, "input is empty"},
    {ACHIP_D_LZ4_MALFORMED, "Malformed input"},
    {ACHIP_D_LZ4_LAST_LITERAL_OUTSIDE, "attempt to write last literal outside of destination buffer"},
    {ACHIP_D_LZ4_INPUT_NOT_CONSUMED, "all input must be consumed"},
    {ACHIP_D_LZ4_OFFSET_OUTSIDE, "offset outside destination buffer"},
    {ACHIP_D_LZ4_LAST_5_LITERALS, "last 5 bytes must be literals"},
    {ACHIP_D_LZ4_EMPTY_OUTPUT, "Output buffer too small"},
    {ACHIP_D_LZ4_MAX_INPUT, "Max input length exceeded"},
    {ACHIP_D_LZ4_MAX_OUTPUT, "Max output length must be larger than the LZ4 bound"},
    {ACHIP_D_LZ4F_TOO_SHORT, "Input is too short to be an LZ4 frame"},
    {ACHIP_D_LZ4F_TRUNC_MAGIC, "Truncated LZ4 frame: incomplete magic number"},
    {ACHIP_D_LZ4F_BAD_MAGIC, "Invalid LZ4 frame magic number"},
    {ACHIP_D_LZ4F_TRUNC_HEADER, "Truncated LZ4 frame header"},
    {ACHIP_D_LZ4F_VERSION_0, "Unsupported LZ4 frame version: 0"},
    {ACHIP_D_LZ4F_VERSION_2, "Unsupported LZ4 frame version: 2"},
    {ACHIP_D_LZ4F_VERSION_3, "Unsupported LZ4 frame version: 3"},
    {ACHIP_D_LZ4F_RESERVED_BITS, "Corrupt LZ4 frame: reserved bits in the frame descriptor must be zero"},
    {ACHIP_D_LZ4F_LINKED_BLOCKS, "LZ4 frames with linked blocks are not supported"},
    {ACHIP_D_LZ4F_DICTIONARY, "LZ4 frames with a dictionary are not supported"},
    {ACHIP_D_LZ4F_BLOCK_MAX_SIZE, "Invalid LZ4 frame block maximum size"},
    {ACHIP_D_LZ4F_HEADER_CHECKSUM, "Corrupt LZ4 frame: invalid header checksum"},
    {ACHIP_D_LZ4F_MISSING_BLOCK_SIZE, "Truncated LZ4 frame: missing block size"},
    {ACHIP_D_LZ4F_BLOCK_PAST_END, "Truncated LZ4 frame: block extends past end of input"},
    {ACHIP_D_LZ4F_OUTPUT_TOO_SMALL, "Output buffer too small"},
    {ACHIP_D_LZ4F_BLOCK_EXCEEDS_MAX, "Corrupt LZ4 frame: decompressed block exceeds maximum block size"},
    {ACHIP_D_LZ4F_MISSING_BLOCK_CHECKSUM, "Truncated LZ4 frame: missing block checksum"},
    {ACHIP_D_LZ4F_BLOCK_CHECKSUM, "Corrupt LZ4 frame: invalid block checksum"},
    {ACHIP_D_LZ4F_MISSING_CONTENT_CHECKSUM, "Truncated LZ4 frame: missing content checksum"},
    {ACHIP_D_LZ4F_CONTENT_CHECKSUM, "Corrupt LZ4 frame: invalid content checksum"},
    {ACHIP_D_LZ4F_CONTENT_SIZE, "Corrupt LZ4 frame: content size does not match frame header"},
    {ACHIP_D_LZ4F_TRUNC_SKIP_SIZE, "Truncated LZ4 skippable frame: missing frame size"},
    {ACHIP_D_LZ4F_TRUNC_SKIP, "Truncated LZ4 skippable frame"},
    {ACHIP_D_LZ4F_MAX_OUTPUT, "Output buffer too small"},
    {ACHIP_D_SNF_EOF_STREAM_HEADER, "encountered EOF while reading stream header"},
    {ACHIP_D_SNF_BAD_STREAM_HEADER, "invalid stream header"},
    {ACHIP_D_SNF_EOF_BLOCK_HEADER, "encountered EOF while reading block header"},
    {ACHIP_D_SNF_EOF_FRAME, "unexpectd EOF when reading frame"},
    {ACHIP_D_SNF_STREAM_ID_LENGTH, "stream identifier chunk with invalid length"},
    {ACHIP_D_SNF_UNSKIPPABLE, "unsupported unskippable chunk"},
    {ACHIP_D_SNF_INVALID_LENGTH, "invalid length for chunk flag"},
    {ACHIP_D_SNF_CHECKSUM, "Corrupt input: invalid checksum"},
    {ACHIP_D_SNF_OUTPUT_TOO_SMALL, "Output buffer too small for the stream"},
    {ACHIP_D_SNF_MAX_OUTPUT, "Output buffer too small"},
    {ACHIP_D_HDP_TRUNCATED_INT, "Stream is truncated"},
    {ACHIP_D_HDP_EOF_BLOCK_DATA, "encountered EOF while reading block data"},
    {ACHIP_D_HDP_CHUNK_EXCEEDS_BLOCK, "Chunk uncompressed size is greater than block size"},
    {ACHIP_D_HDP_LENGTH_MISMATCH, "Expected to read the chunk's announced bytes, but data only contained fewer"},
    {ACHIP_D_HDP_NOT_CONSUMED, "All input was not consumed"},
    {ACHIP_D_HDP_NEGATIVE_LENGTH, "negative chunk length"},
    {ACHIP_D_HDP_MAX_OUTPUT, "Output buffer too small"},
    {ACHIP_D_SNAPPY_MALFORMED, "Malformed input"},
    {ACHIP_D_SNAPPY_TRUNCATED, "Input is truncated"},
    {ACHIP_D_SNAPPY_LEN_HIGH_BIT, "last byte of compressed length int has high bit set"},
    {ACHIP_D_SNAPPY_INVALID_LENGTH, "invalid compressed length"},
    {ACHIP_D_SNAPPY_LENGTH_MISMATCH, "Recorded length differs from actual length after decompression"},
    {ACHIP_D_SNAPPY_OUTPUT_TOO_SMALL, "Uncompressed length must be less than the output buffer size"},
    {ACHIP_D_SNAPPY_MAX_OUTPUT, "Output buffer must be at least the Snappy bound"},
    {ACHIP_D_ZSTD_NOT_ENOUGH_INPUT, "Not enough input bytes"},
    {ACHIP_D_ZSTD_OUTPUT_TOO_SMALL, "Output buffer too small"},
    {ACHIP_D_ZSTD_CORRUPTED, "Input is corrupted"},
    {ACHIP_D_ZSTD_BAD_MAGIC, "Invalid magic prefix"},
    {ACHIP_D_ZSTD_V07_MAGIC, "Data encoded in unsupported ZSTD v0.7 format"},
    {ACHIP_D_ZSTD_BAD_CHECKSUM, "Bad checksum"},
    {ACHIP_D_ZSTD_DICTIONARY, "Custom dictionaries not supported"},
    {ACHIP_D_ZSTD_INVALID_BLOCK_TYPE, "Invalid block type"},
    {ACHIP_D_ZSTD_BLOCK_TOO_LARGE, "Expected match length table to be present"},
    {ACHIP_D_ZSTD_BLOCK_TOO_SMALL, "Compressed block size too small"},
    {ACHIP_D_ZSTD_WINDOW_TOO_LARGE, "Window size too large (not yet supported)"},
    {ACHIP_D_ZSTD_DICT_CORRUPTED, "Dictionary is corrupted"},
    {ACHIP_D_ZSTD_LITERALS_TOO_LARGE, "Block exceeds maximum size"},
    {ACHIP_D_ZSTD_FSE_TABLE_LOG, "FSE table size exceeds maximum allowed size"},
    {ACHIP_D_ZSTD_FSE_SYMBOL, "Symbol larger than max value"},
    {ACHIP_D_ZSTD_TABLE_MISSING, "Expected match length table to be present"},
    {ACHIP_D_ZSTD_VALUE_TOO_LARGE, "Value exceeds expected maximum value"},
    {ACHIP_D_ZSTD_BITSTREAM_EMPTY, "Bitstream is empty"},
    {ACHIP_D_ZSTD_BITSTREAM_NO_MARK, "Bitstream end mark not present"},
    {ACHIP_D_ZSTD_BITSTREAM_NOT_CONSUMED, "Bit stream is not fully consumed"},
    {ACHIP_D_ZSTD_SEQUENCES_NOT_CONSUMED, "Not all sequences were consumed"},
    {ACHIP_D_ZSTD_FSE_OUTPUT_SMALL, "Output buffer is too small"},
    {ACHIP_D_ZSTD_MAX_OUTPUT, "Output buffer too small"},
    {ACHIP_D_NO_DEVICE, "No HIP device available"},
    {ACHIP_D_HIP_ERROR, "HIP runtime error"},
    {ACHIP_D_BAD_ARGUMENT, "Invalid argument"},
    {ACHIP_D_UNSUPPORTED, "Operation not supported by this build"},
};

}  // namespace

extern "C" {

int32_t achip_status_class(int32_t status) { return status < 0 ? ((-status) & 15) : 0; }
int32_t achip_status_detail(int32_t status) { return status < 0 ? ((-status) >> 4) : 0; }

const char* achip_detail_message(int32_t detail)
{
    for (const DetailText& d : kDetailText) {
        if (d.detail == detail) {
            return d.text;
        }
    }
    return "Unknown error";
}

const char* achip_version(void) { return "aircompressor-hip 0.1 (gfx950)"; }

int32_t achip_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

const char* achip_last_error(void) { return g_lastError.c_str(); }

// ---- size helpers -------------------------------------------------------
int32_t achip_lz4_max_compressed_length(int32_t n) { return n + n / 255 + 16; }
int32_t achip_snappy_max_compressed_length(int32_t n) { return 32 + n + n / 6; }
int32_t achip_lz4frame_max_compressed_length(int32_t n)
{
    // Lz4FrameCompression.maxCompressedLength  M/lz4/Lz4FrameCompression.java:70-83
    if (n < 0) return bad_argument("uncompressedSize is negative");
    const int64_t blocks = ((int64_t)n + (4 << 20) - 1) / (4 << 20);
    const int64_t maxLength = 7 + 4 + (int64_t)n + 4 * blocks;
    if (maxLength > 0x7FFFFFFF) return bad_argument("Maximum compressed length exceeds Integer.MAX_VALUE");
    return (int32_t)maxLength;
}
int32_t achip_snappyframed_max_compressed_length(int32_t n)
{
    // stream header + per 64 KiB block a chunk header, the masked CRC and at most the block itself (a compressed chunk is kept
    // only at <= 0.85 of its block: M/snappy/SnappyFramedOutputStream.java:214)
    if (n < 0) return bad_argument("uncompressedSize is negative");
    const int64_t blocks = ((int64_t)n + 65535) / 65536;
    const int64_t maxLength = 10 + 8 * blocks + (int64_t)n;
    if (maxLength > 0x7FFFFFFF) return bad_argument("Maximum compressed length exceeds Integer.MAX_VALUE");
    return (int32_t)maxLength;
}
int32_t achip_hadoop_max_compressed_length(int32_t codec, int32_t n, int32_t bufferSize)
{
    // per chunk of bufferSize - overhead plaintext bytes: two big-endian ints and at most the codec's maxCompressedLength
    // (M/lz4/Lz4HadoopOutputStream.java:44-46, 107-118, 128-131; M/snappy/SnappyHadoopOutputStream.java likewise)
    if (n < 0) return bad_argument("uncompressedSize is negative");
    if (codec != 0 && codec != 1) return bad_argument("codec must be 0 (LZ4) or 1 (Snappy)");
    const bool snappy = codec == 1;
    const int64_t overhead = snappy ? bufferSize / 6 + 32 : ((int32_t)(bufferSize * 0.01) > 10 ? (int32_t)(bufferSize * 0.01) : 10);
    const int64_t chunk = (int64_t)bufferSize - overhead;
    if (bufferSize <= 0 || chunk <= 0) return bad_argument("bufferSize too small");
    auto bound = [&](int64_t m) { return snappy ? 32 + m + m / 6 : m + m / 255 + 16; };
    const int64_t rest = (int64_t)n % chunk;
    const int64_t maxLength = ((int64_t)n / chunk) * (8 + bound(chunk)) + (rest > 0 ? 8 + bound(rest) : 0);
    if (maxLength > 0x7FFFFFFF) return bad_argument("Maximum compressed length exceeds Integer.MAX_VALUE");
    return (int32_t)maxLength;
}
int32_t achip_zstdstream_max_compressed_length(int32_t n)
{
    if (n < 0) return bad_argument("uncompressedSize is negative");
    const int64_t r = (int64_t)achip_zstd_max_compressed_length(n) + 16;
    if (r > 0x7FFFFFFF) return bad_argument("Maximum compressed length exceeds Integer.MAX_VALUE");
    return (int32_t)r;
}
int32_t achip_zstd_max_compressed_length(int32_t n)
{
    int32_t result = n + (int32_t)((uint32_t)n >> 8);
    if (n < 128 * 1024) {
        result += (int32_t)((uint32_t)(128 * 1024 - n) >> 11);
    }
    return result;
}

int64_t achip_snappy_uncompressed_length(const void* src, int64_t srcLen, int64_t* errOffset)
{
    // SnappyRawDecompressor.readUncompressedLength  M/snappy/SnappyRawDecompressor.java:277-321
    const uint8_t* in = (const uint8_t*)src;
    uint32_t result = 0;
    int64_t n = 0;
    for (int i = 0; i < 5; i++) {
        if (n >= srcLen) {
            if (errOffset) *errOffset = srcLen - n;
            return ACHIP_STATUS(ACHIP_CLASS_MALFORMED, ACHIP_D_SNAPPY_TRUNCATED);
        }
        uint32_t b = in[n++];
        result |= (b & 0x7f) << (7 * i);
        if ((b & 0x80) == 0) {
            break;
        }
        if (i == 4) {
            if (errOffset) *errOffset = n;
            return ACHIP_STATUS(ACHIP_CLASS_MALFORMED, ACHIP_D_SNAPPY_LEN_HIGH_BIT);
        }
    }
    if ((int32_t)result < 0) {
        if (errOffset) *errOffset = 0;
        return ACHIP_STATUS(ACHIP_CLASS_MALFORMED, ACHIP_D_SNAPPY_INVALID_LENGTH);
    }
    return (int64_t)result;
}

// An upper bound of what the frames in [src, src + srcLen) decode to -- what a one-shot decoder needs before it can read a stream
// whose frames carry NO content size (ZstdOutputStream writes such frames from 4 MiB on, M/zstd/ZstdOutputStream.java:193-221; the
// reference reads them through a growing window, M/zstd/ZstdIncrementalFrameDecompressor.java:99-234,305-352, never knowing the size).
// Walks the frame headers (readFrameHeader, M/zstd/ZstdFrameDecompressor.java:865-947) and the block headers (:156-181): a raw or RLE
// block decodes to its size field, a compressed block to at most MAX_BLOCK_SIZE = 128 KiB (:278), a frame to at most its content
// size when it has one.  Host code, no device.  Negative = status (the bytes do not parse as frames; *errOffset set).
int64_t achip_zstd_decompress_bound(const void* src, int64_t srcLen, int64_t* errOffset)
{
    const uint8_t* in = (const uint8_t*)src;
    auto fail = [&](int detail, int64_t off) -> int64_t {
        if (errOffset) *errOffset = off;
        return ACHIP_STATUS(ACHIP_CLASS_MALFORMED, detail);
    };
    if (errOffset) *errOffset = 0;
    if (srcLen < 0 || (srcLen > 0 && in == nullptr)) return bad_argument("src");
    int64_t input = 0, total = 0;
    while (input < srcLen) {
        int64_t eo = 0;
        const int64_t fcs = achip_zstd_decompressed_size(in + input, srcLen - input, &eo);
        if (fcs < -1) {
            if (errOffset) *errOffset = input + eo;
            return fcs;
        }
        const int32_t fhd = in[input + 4];
        const bool singleSegment = (fhd & 0x20) != 0, hasChecksum = (fhd & 0x04) != 0;
        const int32_t csDesc = fhd >> 6;
        input += 4 + 1 + (singleSegment ? 0 : 1) + (csDesc == 0 ? (singleSegment ? 1 : 0) : (1 << csDesc));
        int64_t blocks = 0;
        for (;;) {
            if (srcLen - input < 3) return fail(ACHIP_D_ZSTD_NOT_ENOUGH_INPUT, input);
            const int32_t h = in[input] | (in[input + 1] << 8) | (in[input + 2] << 16);
            input += 3;
            const int32_t type = (h >> 1) & 3, size = h >> 3;
            if (type == 3) return fail(ACHIP_D_ZSTD_INVALID_BLOCK_TYPE, input);
            const int64_t stored = type == 1 ? 1 : size;  // an RLE block stores one byte
            if (stored > srcLen - input) return fail(ACHIP_D_ZSTD_NOT_ENOUGH_INPUT, input);
            input += stored;
            blocks += type == 2 ? 131072 : size;
            if (h & 1) {
                break;
            }
        }
        if (hasChecksum) {
            if (srcLen - input < 4) return fail(ACHIP_D_ZSTD_NOT_ENOUGH_INPUT, input);
            input += 4;
        }
        total += fcs >= 0 && fcs < blocks ? fcs : blocks;
    }
    return total;
}

int64_t achip_zstd_decompressed_size(const void* src, int64_t srcLen, int64_t* errOffset)
{
    // ZstdFrameDecompressor.getDecompressedSize = verifyMagic + readFrameHeader
    // M/zstd/ZstdFrameDecompressor.java:860-962
    const uint8_t* in = (const uint8_t*)src;
    auto fail = [&](int detail, int64_t off) -> int64_t {
        if (errOffset) *errOffset = off;
        return ACHIP_STATUS(ACHIP_CLASS_MALFORMED, detail);
    };
    auto rd = [&](int64_t pos, int n) -> uint64_t {
        uint64_t v = 0;
        for (int i = 0; i < n; i++) {
            if (pos + i < srcLen) v |= (uint64_t)in[pos + i] << (8 * i);
        }
        return v;
    };
    if (srcLen < 4) return fail(ACHIP_D_ZSTD_NOT_ENOUGH_INPUT, 0);
    uint32_t magic = (uint32_t)rd(0, 4);
    if (magic != 0xFD2FB528u) {
        return fail(magic == 0xFD2FB527u ? ACHIP_D_ZSTD_V07_MAGIC : ACHIP_D_ZSTD_BAD_MAGIC, 0);
    }
    int64_t input = 4;
    if (!(input < srcLen)) return fail(ACHIP_D_ZSTD_NOT_ENOUGH_INPUT, input);
    int32_t fhd = (int32_t)rd(input++, 1);
    bool singleSegment = (fhd & 0x20) != 0;
    int32_t dictDesc = fhd & 3;
    int32_t csDesc = fhd >> 6;
    int32_t headerSize = 1 + (singleSegment ? 0 : 1) + (dictDesc == 0 ? 0 : (1 << (dictDesc - 1))) +
                         (csDesc == 0 ? (singleSegment ? 1 : 0) : (1 << csDesc));
    if (headerSize > srcLen - 4) return fail(ACHIP_D_ZSTD_NOT_ENOUGH_INPUT, input);
    if (!singleSegment) input++;
    if (dictDesc != 0) return fail(ACHIP_D_ZSTD_DICTIONARY, input + (1 << (dictDesc - 1)));
    switch (csDesc) {
        case 0: return singleSegment ? (int64_t)rd(input, 1) : -1;
        case 1: return (int64_t)rd(input, 2) + 256;
        case 2: return (int64_t)rd(input, 4);
        default: {
            // the reference returns the raw long; a field >= 2^63 would collide with this API's negative statuses, so it is
            // reported as what it is (no such frame can be decoded: the window check rejects it)
            const uint64_t v = rd(input, 8);
            return v > (uint64_t)INT64_MAX ? fail(ACHIP_D_ZSTD_WINDOW_TOO_LARGE, input) : (int64_t)v;
        }
    }
}

// ---- context -----------------------------------------------------------
achip_ctx* achip_ctx_create(int32_t device)
{
    int n = achip_device_count();
    if (n <= 0 || device < 0 || device >= n) {
        g_lastError = n <= 0 ? "no HIP device" : "device ordinal out of range";
        return nullptr;
    }
    if (hipSetDevice(device) != hipSuccess) {
        g_lastError = "hipSetDevice failed";
        return nullptr;
    }
    achip_ctx* ctx = new achip_ctx();
    ctx->device = device;
    hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        device_failure("hipStreamCreate", e);
        delete ctx;
        return nullptr;
    }
    return ctx;
}

void achip_ctx_destroy(achip_ctx* ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    if (ctx->zstdMbScratch) (void)hipFree(ctx->zstdMbScratch);
    if (ctx->hostStage) (void)hipHostFree(ctx->hostStage);
    if (ctx->devStage) (void)hipFree(ctx->devStage);
    if (ctx->mixHost) (void)hipHostFree(ctx->mixHost);
    if (ctx->mixDev) (void)hipFree(ctx->mixDev);
    if (ctx->mixUploaded) (void)hipEventDestroy(ctx->mixUploaded);
    if (ctx->mixGathered) (void)hipEventDestroy(ctx->mixGathered);
    if (ctx->autoEv) (void)hipEventDestroy(ctx->autoEv);
    if (ctx->autoPinned) (void)hipHostFree(ctx->autoPinned);
    for (int k = 0; k < 3; k++) {
        if (ctx->mixLaneDone[k]) (void)hipEventDestroy(ctx->mixLaneDone[k]);
        if (ctx->mixLane[k]) achip_ctx_destroy(ctx->mixLane[k]);
    }
    destroy_host_path(ctx);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int32_t achip_ctx_device(achip_ctx* ctx) { return ctx ? ctx->device : -1; }
void* achip_ctx_stream(achip_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int32_t achip_ctx_synchronize(achip_ctx* ctx)
{
    if (!ctx) return bad_argument("ctx is null");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

int32_t achip_ctx_set_option(achip_ctx* ctx, const char* name, int64_t value)
{
    if (!ctx || !name) return bad_argument("ctx/name is null");
    std::string k(name);
    auto pow2 = [](int64_t v) { return v >= 1 && v <= 64 && (v & (v - 1)) == 0; };
    if (k == "lz4.decompress.group") {
        if (!pow2(value)) return bad_argument("group size must be a power of two in 1..64");
        ctx->lz4dGroup = (int)value;
    }
    else if (k == "snappy.decompress.group") {
        if (value != 0 && !pow2(value)) return bad_argument("snappy.decompress.group: 0 (by the batch size) or a power of two in 1..64");
        ctx->snappydGroup = (int)value;
    }
    else if (k == "lz4.decompress.variant") {
        if (value != 1 && value != 5 && value != 7) return bad_argument("lz4.decompress.variant: 1 rings, 7 two passes, 5 auto");
        ctx->lz4dVariant = (int)value;
    }
    else if (k == "lz4.decompress.auto_min_blocks") {
        if (value < 16 || value > 0x7FFFFFFF) return bad_argument("lz4.decompress.auto_min_blocks must be at least 16");
        ctx->lz4dAutoMinBlocks = (int)value;
    }
    else if (k == "snappy.decompress.variant") {
        if (value != 1 && value != 5 && value != 7) return bad_argument("snappy.decompress.variant: 1 rings, 7 two passes, 5 auto");
        ctx->snappydVariant = (int)value;
    }
    else if (k == "decompress.ring_class") {
        if (value < 0 || value > 2) return bad_argument("decompress.ring_class: 0 compact (4 lanes per block: phased), 1 large, 2 round-2 compact rings (4 lanes per block)");
        ctx->ringClass = (int)value;
    }
    else if (k == "lz4.compress.variant") {
        if (value != 0 && value != 1 && value != 4) return bad_argument("lz4.compress.variant: 0 serial probes, 1 batch probes, 4 many matches per window");
        ctx->lz4cVariant = (int)value;
    }
    else if (k == "snappy.compress.variant") {
        if (value < 0 || value > 4 || value == 3) return bad_argument("snappy.compress.variant: 0 serial probes, 1 batch probes, 2 two tiers, 4 two tiers, many matches per window");
        ctx->snappycVariant = (int)value;
    }
    else if (k == "snappyframed.decompress.variant") {
        if (value < 0 || value > 3) return bad_argument("snappyframed.decompress.variant: 0 a wavefront per stream, 1 chunk list through the ring decoders, 2 through the two-pass decoder, 3 chosen by a probe");
        ctx->snappyFramedVariant = (int)value;
    }
    else if (k == "snappyframed.compress.variant") {
        if (value != 0 && value != 1) return bad_argument("snappyframed.compress.variant: 0 a wavefront per stream, 1 block list");
        ctx->snappyFramedCompressVariant = (int)value;
    }
    else if (k == "hadoop.buffer_size") {
        if (value < 64 || value > 0x40000000) return bad_argument("hadoop.buffer_size out of range");
        ctx->hadoopBufferSize = (int)value;
    }
    else if (k == "hadoop.decompress.variant") {
        if (value < 0 || value > 3) return bad_argument("hadoop.decompress.variant: 0 a wavefront per stream, 1 chunk list through the ring decoders, 2 through the two-pass decoders, 3 chosen by a probe");
        ctx->hadoopDecompressVariant = (int)value;
    }
    else if (k == "lz4frame.decompress.variant") {
        if (value < 0 || value > 2) return bad_argument("lz4frame.decompress.variant: 0 a wavefront per item, 1 block list through the two-pass decoder, 2 chosen by a probe");
        ctx->lz4FrameDecompressVariant = (int)value;
    }
    else if (k == "lz4.decompress.parse") {
        if (value < 0 || value > 2) return bad_argument("lz4.decompress.parse: 0 by the batch (a wavefront per block below 32768 blocks), 1 a lane per block, 2 a wavefront per block");
        achip::g_lz4_parse_mode = (int)value;
    }
    else if (k == "mixed.concurrent") {
        if (value != 0 && value != 1) return bad_argument("mixed.concurrent: 1 a mixed batch's codec families side by side (a stream and scratch each), 0 every bucket in turn");
        ctx->mixConcurrent = (int)value;
    }
    else if (k == "snappy.decompress.parse") {
        if (value < 0 || value > 2) return bad_argument("snappy.decompress.parse: 0 by the batch (a wavefront per block up to 4096 blocks), 1 a lane per block, 2 a wavefront per block");
        achip::g_snappy_parse_mode = (int)value;
    }
    else if (k == "zstd.decompress.exec") {
        if (value < 0 || value > 2) return bad_argument("zstd.decompress.exec: 0 rings, 1 record executor, 2 chosen per item");
        achip::g_zstd_pipe_exec = (int)value;
    }  // (process-wide: a development switch between the two execute stages)
    else if (k == "zstd.decompress.seq_waves") {
        if (value != 1 && value != 2 && value != 4) return bad_argument("zstd.decompress.seq_waves: wavefronts per workgroup of the pipeline's sequence stage: 1, 2 or 4 (64 items a workgroup either way)");
        achip::g_zstd_seq_waves = (int)value;
    }  // (process-wide, like zstd.decompress.exec)
    else if (k == "zstd.decompress.lit_items") {
        if (value != 8 && value != 10 && value != 13 && value != 16 && value != 20) return bad_argument("zstd.decompress.lit_items: items per wavefront of the pipeline's literal stage: 8, 10 or 16 (4 KiB of LDS an item), 13 (3 KiB: symbols and length nibbles apart), 20 (16 items of 2 304 bytes: symbols, and lengths by symbol)");
        achip::g_zstd_lit_items = (int)value;
    }  // (process-wide)
    else if (k == "decompress.latency_max_blocks") {
        if (value < 0 || value > 65536) return bad_argument("decompress.latency_max_blocks: 0 (never) .. 65536: LZ4 / Snappy batches of at most this many blocks take a wavefront and 128 KiB of LDS history per block");
        ctx->latencyMaxBlocks = (int)value;
    }
    else if (k == "decompress.ring_pad") {
        if (value < 0 || value > 256 || (value & 15) != 0) return bad_argument("ring pad must be a multiple of 16 in 0..256");
        ctx->ringPad = (int)value;
    }
    else if (k == "zstd.decompress.tile") {
        if (value < 64 || value > 65536) return bad_argument("tile must be in 64..65536");
        ctx->zstdTile = (int)value;
    }
    else if (k == "debug.scratch_poison") ctx->scratchPoison = (int)value;
    else if (k == "zstd.decompress.variant") {
        if (value != 0 && value != 1) return bad_argument("zstd.decompress.variant: 1 pipeline, 0 one-kernel decoder");
        ctx->zstddVariant = (int)value;
    }
    else if (k == "zstd.stream.chunked") ctx->zstdStreamChunked = value != 0 ? 1 : 0;
    else if (k == "zstd.decompress.stream_blocks") {
        if (value != 0 && (value < 16 || value > 131072)) return bad_argument("zstd.decompress.stream_blocks must be 0 or 16..131072");
        ctx->zstdStreamBlocks = (int)value;
    }
    else if (k == "zstd.compress.variant") {
        const bool ok = value >= 0 && value <= 3;
        if (!ok) return bad_argument("zstd.compress.variant: 3 match-finder kernel (many matches per window) + entropy kernel, 0 the same with batch probes, 1 with serial probes, 2 one kernel");
        ctx->zstdcVariant = (int)value;
    }
    else if (k == "host.look_max_blocks") {
        if (value < 0 || value > 65536) return bad_argument("host.look_max_blocks: 0 .. 65536");
        ctx->hostLookMaxBlocks = (int)value;
    }
    else if (k == "host.copy_priority") {
        if (value != 0 && value != 1) return bad_argument("host.copy_priority: 1 the host-pointer pipeline's copy streams at the lowest priority (default), 0 at the default priority");
        ctx->hostCopyLowPriority = (int)value;
    }
    else if (k == "decompress.auto_remember") {
        if (value != 0 && value != 1) return bad_argument("decompress.auto_remember: 1 auto mode launches only the decoder the last arrived probe statistics chose for batches of that shape (default), 0 both decoders in every call");
        ctx->autoRemember = (int)value;
        ctx->autoChoice[0] = ctx->autoChoice[1] = -1;
    }
    else if (k == "host.ramp") {
        if (value != 0 && value != 1) return bad_argument("host.ramp: 1 smaller chunks at the start and the end of a host-pointer batch (default), 0 equal chunks");
        ctx->hostRamp = (int)value;
    }
    else if (k == "host.slots") {
        if (value < 2 || value > achip_ctx::kHostSlots) return bad_argument("host.slots: 2 .. 8 staging slots of the host-pointer pipeline");
        ctx->hostSlots = (int)value;
    }
    else if (k == "host.blit") {
        if (value < 0 || value > 3) return bad_argument("host.blit: bit 0 = the host-pointer pipeline's uploads by a copy kernel, bit 1 = its downloads (0 = both by hipMemcpyAsync)");
        ctx->hostBlit = (int)value;
    }
    else if (k == "host.blit_groups") {
        if (value < 1 || value > 4096) return bad_argument("host.blit_groups: 1 .. 4096 workgroups of the copy kernel");
        ctx->hostBlitGroups = (int)value;
    }
    else if (k == "max_src_len_hint") ctx->maxSrcLenHint = (int)value;
    else if (k == "lz4.compress.mem_waves") {
        if (value < 0 || value > 2) return bad_argument("lz4.compress.mem_waves: wavefronts per workgroup of the window encoder whose table lies in memory: 0 (one wavefront per block, table in LDS), 1 or 2");
        achip::g_lz4_mem_waves = (int)value;
    }  // (process-wide)
    else if (k == "lz4.compress.tier_min_blocks") {
        if (value < 1 || value > (1 << 30)) return bad_argument("lz4.compress.tier_min_blocks: batches of at least this many blocks take the two-tier kernel (default 5120)");
        achip::g_lz4_tier_min_blocks = (int)value;
    }  // (process-wide)
    else if (k == "snappy.compress.mem_waves") {
        if (value < 0 || value > 3) return bad_argument("snappy.compress.mem_waves: wavefronts per workgroup of the two-tier encoder whose table lies in memory, 0 .. 3");
        achip::g_snappy_mem_waves = (int)value;
    }  // (process-wide)
    else if (k == "snappy.compress.fan") {
        if (value != 0 && value != 1) return bad_argument("snappy.compress.fan: 1 the independent 64 KiB sub-blocks of a buffer side by side (default), 0 in turn on one wavefront");
        ctx->snappyFan = (int)value;
    }
    else if (k == "decompress.exec_variant") {
        // 2: the one executor there is.  (Round 2's experiments and timing aids -- 121 .. 125, 201, 302 .. 308 -- were measured, then removed: rounds 3 and 4.)
        const bool ok = value == 2;
        if (!ok) return bad_argument("decompress.exec_variant: 2");
        ctx->execVariant = (int)value;
    }
    else if (k == "host.chunk_bytes") {
        if (value < (1 << 16) || value > (1LL << 32)) return bad_argument("host.chunk_bytes must be in 64 KiB .. 4 GiB");
        ctx->hostChunkBytes = value;
    }
    else if (k == "host.copy_threads") {
        if (value < 0 || value > 64) return bad_argument("host.copy_threads must be in 0..64");
        if (ctx->pool) return bad_argument("host.copy_threads must be set before the first host-pointer batch");
        ctx->hostCopyThreads = (int)value;
    }
    else return bad_argument("unknown option");
    return 0;
}

int64_t achip_ctx_get_stat(achip_ctx* ctx, const char* name)
{
    if (!ctx || !name) return -1;
    std::string k(name);
    if (k == "lz4.decompress.mixed_groups") {  // auto mode's probe result of the last LZ4 decode (-1: it did not run)
        if (!ctx->lastLz4dAuto || ctx->scratch == nullptr) return -1;
        if (hipSetDevice(ctx->device) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) return -1;
        int32_t v = 0;
        if (hipMemcpy(&v, ctx->scratch, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return -1;
        return v;
    }
    if (k == "host.gather_us") return ctx->hostGatherUs;          // the gather thread copying the caller's inputs into pinned slots
    if (k == "host.scatter_us") return ctx->hostScatterUs;        // the finalizer thread copying outputs to the caller's buffers
    if (k == "host.wait_slot_us") return ctx->hostWaitSlotUs;     // the gather thread waiting for a free slot (the pipeline behind it is the limit)
    if (k == "host.wait_download_us") return ctx->hostWaitDownloadUs;  // the finalizer waiting for a chunk's download (the device side / the gather is the limit)
    if (k == "host.chunks") return ctx->hostChunks;
    if (k == "host.total_us") return ctx->hostTotalUs;
    if (k == "decompress.choice") {  // which decoder auto mode ran last: 0 rings, 3 two passes; -1: no probe ran
        if (ctx->lastRemembered >= 0) return ctx->lastRemembered;  // (a remembered choice: decompress.auto_reprobe)
        if (!ctx->lastLz4dAuto || ctx->scratch == nullptr) return -1;
        if (hipSetDevice(ctx->device) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) return -1;
        int32_t v[6] = {0, 0, 0, 0, 0, 0};
        if (hipMemcpy(v, ctx->scratch, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return -1;
        const bool mixed = (int64_t)v[0] * 4 > (ctx->lastAutoBlocks + 15) / 16;  // the rule of lz4_pick (achip_device.h)
        const bool pooledShort = v[1] > 0 && (int64_t)v[2] < (ctx->lastAutoIsLz4 ? 12 : 6) * (int64_t)v[1];
        const bool isShort = v[5] > 0 ? (int64_t)v[4] * 3 > (int64_t)v[5] : pooledShort;
        return (mixed || isShort) ? 3 : 0;
    }
    if (k == "decompress.scratch_bytes") {  // the context's decode scratch as granted (the two-pass decoders' record arena is what lies behind its fixed part): a smaller grant than a batch asked for shows here and in decompress.twopass_fallback_blocks
        return ctx->scratchBytes;
    }
    if (k == "decompress.twopass_fallback_blocks") {  // blocks the last two-pass LZ4 / Snappy decode handed to the ring decoder (-1: none ran)
        if (!ctx->lastTwopass || ctx->scratch == nullptr) return -1;
        if (hipSetDevice(ctx->device) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) return -1;
        int32_t v[3] = {0, 0, 0};
        if (hipMemcpy(v, (const uint8_t*)ctx->scratch + (ctx->lastLz4dAuto ? 4096 : 0), sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return -1;
        return v[2];
    }
    if (k == "zstd.decompress.multiblock_items" || k == "zstd.decompress.multiblock_blocks" || k == "zstd.decompress.multiblock_fast_items") {
        // the last Zstd decode: items K1 handed to the multi-block stages, their blocks, items those stages finished
        if (ctx->lastZstddBlocks <= 0 || ctx->scratch == nullptr || ctx->lastZstddVariant == 0) return -1;
        if (hipSetDevice(ctx->device) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) return -1;
        int32_t v = 0;
        const int word = k == "zstd.decompress.multiblock_items" ? 40 : (k == "zstd.decompress.multiblock_blocks" ? 41 : 42);
        if (hipMemcpy(&v, (const int32_t*)ctx->scratch + word, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return -1;
        return v;
    }
    const std::string prefix = "zstd.decompress.fallback_";
    if (k.compare(0, prefix.size(), prefix) == 0) {
        // "items": all items handed to the one-kernel decoder; "stage1".."stage5": by the stage that handed them over
        const std::string what = k.substr(prefix.size());
        int word = -1;
        if (what == "items") word = 0;
        else if (what.size() == 6 && what.compare(0, 5, "stage") == 0 && what[5] >= '1' && what[5] <= '6') word = 32 + (what[5] - '0');  // (6: the multi-block stages' walk)
        if (word < 0) return -1;
        if (ctx->lastZstddBlocks <= 0 || ctx->scratch == nullptr) return -1;
        if (ctx->lastZstddVariant == 0) return word == 0 ? ctx->lastZstddBlocks : 0;
        if (hipSetDevice(ctx->device) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) return -1;
        int32_t v = 0;
        if (hipMemcpy(&v, (const int32_t*)ctx->scratch + word, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return -1;  // the pipeline's counters lead its scratch
        return v;
    }
    return -1;
}

// ---- memory helpers ----------------------------------------------------
void* achip_device_alloc(achip_ctx* ctx, int64_t bytes)
{
    if (!ctx || bytes < 0) return nullptr;
    if (hipSetDevice(ctx->device) != hipSuccess) return nullptr;
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, (size_t)std::max<int64_t>(bytes, 1));
    if (e != hipSuccess) {
        device_failure("hipMalloc", e);
        return nullptr;
    }
    return p;
}

int32_t achip_device_free(achip_ctx* ctx, void* p)
{
    if (!ctx) return bad_argument("ctx is null");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipFree(p));
    return 0;
}

void* achip_host_alloc_pinned(int64_t bytes)
{
    void* p = nullptr;
    hipError_t e = hipHostMalloc(&p, (size_t)std::max<int64_t>(bytes, 1), hipHostMallocDefault);
    if (e != hipSuccess) {
        device_failure("hipHostMalloc", e);
        return nullptr;
    }
    return p;
}

int32_t achip_host_free_pinned(void* p)
{
    HIP_TRY(hipHostFree(p));
    return 0;
}

int32_t achip_memcpy_h2d(achip_ctx* ctx, void* dst, const void* src, int64_t bytes)
{
    if (!ctx) return bad_argument("ctx is null");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyHostToDevice, ctx->stream));
    return 0;
}

int32_t achip_memcpy_d2h(achip_ctx* ctx, void* dst, const void* src, int64_t bytes)
{
    if (!ctx) return bad_argument("ctx is null");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToHost, ctx->stream));
    return 0;
}

int32_t achip_memset_d(achip_ctx* ctx, void* dst, int32_t value, int64_t bytes)
{
    if (!ctx) return bad_argument("ctx is null");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemsetAsync(dst, value, (size_t)bytes, ctx->stream));
    return 0;
}

// ---- events ------------------------------------------------------------
void* achip_event_create(void)
{
    hipEvent_t ev;
    if (hipEventCreate(&ev) != hipSuccess) return nullptr;
    return (void*)ev;
}
int32_t achip_event_destroy(void* ev)
{
    HIP_TRY(hipEventDestroy((hipEvent_t)ev));
    return 0;
}
int32_t achip_event_record(achip_ctx* ctx, void* ev)
{
    if (!ctx) return bad_argument("ctx is null");
    HIP_TRY(hipEventRecord((hipEvent_t)ev, ctx->stream));
    return 0;
}
float achip_event_elapsed_ms(void* evStart, void* evStop)
{
    if (hipEventSynchronize((hipEvent_t)evStop) != hipSuccess) return -1.0f;
    float ms = -1.0f;
    if (hipEventElapsedTime(&ms, (hipEvent_t)evStart, (hipEvent_t)evStop) != hipSuccess) return -1.0f;
    return ms;
}

// ---- batched device-resident API ----------------------------------------
#define ACHIP_DEFINE_BATCH(fn, op)                                                                                      \
    int32_t fn(ACHIP_BATCH_ARGS)                                                                                        \
    {                                                                                                                   \
        return launch_op(op, ctx, make_args(srcBase, srcOff, srcLen, dstBase, dstOff, dstCap, outLen, status, errOffset, nBlocks)); \
    }
ACHIP_DEFINE_BATCH(achip_lz4_decompress_batch, ACHIP_OP_LZ4_DECOMPRESS)
ACHIP_DEFINE_BATCH(achip_lz4_compress_batch, ACHIP_OP_LZ4_COMPRESS)
ACHIP_DEFINE_BATCH(achip_snappy_decompress_batch, ACHIP_OP_SNAPPY_DECOMPRESS)
ACHIP_DEFINE_BATCH(achip_snappy_compress_batch, ACHIP_OP_SNAPPY_COMPRESS)
ACHIP_DEFINE_BATCH(achip_zstd_decompress_batch, ACHIP_OP_ZSTD_DECOMPRESS)
ACHIP_DEFINE_BATCH(achip_zstd_compress_batch, ACHIP_OP_ZSTD_COMPRESS)
ACHIP_DEFINE_BATCH(achip_lz4frame_decompress_batch, ACHIP_OP_LZ4FRAME_DECOMPRESS)
ACHIP_DEFINE_BATCH(achip_lz4frame_compress_batch, ACHIP_OP_LZ4FRAME_COMPRESS)
ACHIP_DEFINE_BATCH(achip_snappyframed_decompress_batch, ACHIP_OP_SNAPPYFRAMED_DECOMPRESS)
ACHIP_DEFINE_BATCH(achip_snappyframed_compress_batch, ACHIP_OP_SNAPPYFRAMED_COMPRESS)
ACHIP_DEFINE_BATCH(achip_lz4hadoop_decompress_batch, ACHIP_OP_LZ4HADOOP_DECOMPRESS)
ACHIP_DEFINE_BATCH(achip_lz4hadoop_compress_batch, ACHIP_OP_LZ4HADOOP_COMPRESS)
ACHIP_DEFINE_BATCH(achip_snappyhadoop_decompress_batch, ACHIP_OP_SNAPPYHADOOP_DECOMPRESS)
ACHIP_DEFINE_BATCH(achip_snappyhadoop_compress_batch, ACHIP_OP_SNAPPYHADOOP_COMPRESS)
ACHIP_DEFINE_BATCH(achip_zstdstream_compress_batch, ACHIP_OP_ZSTDSTREAM_COMPRESS)

// ---- mixed batch: bucket by codec op, run every op over its slice, un-bucket (BASELINE configs[4]) ----
namespace {
int32_t ensure_mix(achip_ctx* ctx, int64_t n)
{
    if (n <= ctx->mixItems) return 0;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (ctx->mixHost) { HIP_TRY(hipHostFree(ctx->mixHost)); ctx->mixHost = nullptr; }
    if (ctx->mixDev) { HIP_TRY(hipFree(ctx->mixDev)); ctx->mixDev = nullptr; }
    ctx->mixItems = 0;
    const int64_t want = std::max<int64_t>(n, 4096);
    HIP_TRY(hipHostMalloc((void**)&ctx->mixHost, (size_t)(want * 4), hipHostMallocDefault));
    HIP_TRY(hipMalloc((void**)&ctx->mixDev, (size_t)(want * 48 + 256)));
    if (!ctx->mixUploaded) HIP_TRY(hipEventCreateWithFlags(&ctx->mixUploaded, hipEventDisableTiming));
    ctx->mixItems = want;
    return 0;
}
constexpr int kNumOps = 15;
constexpr int kMixFamilies = 3;
// LZ4 (block, frame, Hadoop), Snappy (block, framed, Hadoop), Zstd (frame, stream): aircompressor_hip.h's ACHIP_OP_* in pairs
int op_family(int op)
{
    static const int family[8] = {0, 1, 2, 0, 1, 0, 1, 2};  // LZ4, Snappy, Zstd, LZ4 frame, x-snappy-framed, LZ4 Hadoop, Snappy Hadoop, Zstd stream
    return family[(op >> 1) & 7];
}
bool op_is_encoder(int op) { return (op & 1) != 0 || op == ACHIP_OP_ZSTDSTREAM_COMPRESS; }  // (aircompressor_hip.h: ACHIP_OP_*_COMPRESS are the odd ops, and the last one)
// the helper context codec family `op` (1, 2) of a mixed batch runs on: made at first use, this context's options at every use
int32_t mix_lane(achip_ctx* ctx, int op, achip_ctx** out)
{
    if (!ctx->mixLane[op]) {
        achip_ctx* lane = achip_ctx_create(ctx->device);
        if (!lane) return ACHIP_STATUS(ACHIP_CLASS_DEVICE, ACHIP_D_HIP_ERROR);
        ctx->mixLane[op] = lane;
    }
    if (!ctx->mixLaneDone[op]) {  // (on its own: a context whose event could not be made at the first attempt tries again, it does not record on a null event)
        HIP_TRY(hipEventCreateWithFlags(&ctx->mixLaneDone[op], hipEventDisableTiming));
    }
    static_cast<achip_options&>(*ctx->mixLane[op]) = static_cast<const achip_options&>(*ctx);
    *out = ctx->mixLane[op];
    return 0;
}
}  // namespace

int32_t achip_mixed_batch(achip_ctx* ctx, const int32_t* codecOp, const void* srcBase, const int64_t* srcOff, const int32_t* srcLen, void* dstBase,
                          const int64_t* dstOff, const int32_t* dstCap, int32_t* outLen, int32_t* status, int64_t* errOffset, int32_t nBlocks)
{
    if (!ctx) return bad_argument("ctx is null");
    if (nBlocks < 0) return bad_argument("nBlocks < 0");
    if (nBlocks == 0) return 0;
    if (!codecOp || !srcOff || !srcLen || !dstOff || !dstCap || !outLen || !status) return bad_argument("null metadata array");
    const int64_t n = nBlocks;
    int64_t count[kNumOps + 1] = {0};
    for (int64_t i = 0; i < n; i++) {
        if (codecOp[i] < 0 || codecOp[i] >= kNumOps) return bad_argument("codecOp out of range");
        count[codecOp[i] + 1]++;
    }
    int32_t r = ensure_mix(ctx, n);
    if (r < 0) return r;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipEventSynchronize(ctx->mixUploaded));  // the previous call's upload has left the pinned buffer (no-op on the first call)
    int64_t start[kNumOps + 1];
    start[0] = 0;
    for (int k = 0; k < kNumOps; k++) start[k + 1] = start[k] + count[k + 1];
    {
        int64_t fill[kNumOps];
        for (int k = 0; k < kNumOps; k++) fill[k] = start[k];
        for (int64_t i = 0; i < n; i++) ctx->mixHost[fill[codecOp[i]]++] = (int32_t)i;  // stable: items of one codec keep their order
    }
    const int64_t cap = ctx->mixItems;
    uint8_t* d = ctx->mixDev;
    int64_t* gSrcOff = (int64_t*)d;
    int64_t* gDstOff = gSrcOff + cap;
    int64_t* gErr = gDstOff + cap;
    int32_t* gSrcLen = (int32_t*)(gErr + cap);
    int32_t* gDstCap = gSrcLen + cap;
    int32_t* gOutLen = gDstCap + cap;
    int32_t* gStatus = gOutLen + cap;
    int32_t* perm = gStatus + cap;
    HIP_TRY(hipMemcpyAsync(perm, ctx->mixHost, (size_t)(n * 4), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipEventRecord(ctx->mixUploaded, ctx->stream));
    const achip::BatchArgs all = make_args(srcBase, srcOff, srcLen, dstBase, dstOff, dstCap, outLen, status, errOffset, nBlocks);
    HIP_TRY(achip::launch_mix_gather(perm, nBlocks, all, gSrcOff, gSrcLen, gDstOff, gDstCap, ctx->stream));
    // The buckets of one codec family in turn (its encoders first: theirs are the long chains -- a file as ONE block is a single wavefront's work for 0.4 s --, and they are
    // launched without a host round trip), the families side by side: LZ4's on this context's stream, Snappy's and Zstd's on a helper context each.
    bool familyUsed[kMixFamilies] = {};
    for (int k = 0; k < kNumOps; k++) familyUsed[op_family(k)] |= count[k + 1] != 0;
    const bool sideBySide = ctx->mixConcurrent != 0 && (int)familyUsed[0] + (int)familyUsed[1] + (int)familyUsed[2] > 1;
    achip_ctx* lanes[kMixFamilies] = {ctx, ctx, ctx};
    bool laneJoined[kMixFamilies] = {};  // helper lanes that were made to wait for the gather (and may have been given work): they must be joined back, also on failure
    // joins the helper lanes used so far back into the context's stream -- on every way out, so that achip_ctx_synchronize covers them and the next
    // call cannot reuse the mixed-batch scratch (gOutLen, gStatus, perm) under kernels of this one
    auto join_lanes = [&]() -> int32_t {
        int32_t jr = 0;
        for (int f = 1; f < kMixFamilies; f++) {
            if (!laneJoined[f]) continue;
            laneJoined[f] = false;
            hipError_t e = hipEventRecord(ctx->mixLaneDone[f], lanes[f]->stream);
            if (e == hipSuccess) e = hipStreamWaitEvent(ctx->stream, ctx->mixLaneDone[f], 0);
            if (e != hipSuccess) {
                (void)hipStreamSynchronize(lanes[f]->stream);  // (the event path failed: wait here instead)
                jr = device_failure("joining the mixed batch's helper lanes", e);
            }
        }
        return jr;
    };
    r = 0;
    if (sideBySide) {
        if (!ctx->mixGathered) HIP_TRY(hipEventCreateWithFlags(&ctx->mixGathered, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(ctx->mixGathered, ctx->stream));
        for (int f = 1; f < kMixFamilies && r >= 0; f++) {
            if (!familyUsed[f]) continue;
            r = mix_lane(ctx, f, &lanes[f]);
            if (r < 0) break;
            const hipError_t e = hipStreamWaitEvent(lanes[f]->stream, ctx->mixGathered, 0);
            if (e != hipSuccess) {
                r = device_failure("hipStreamWaitEvent(lane, gathered)", e);
                break;
            }
            laneJoined[f] = true;
        }
    }
    for (int pass = 0; pass < 2 && r >= 0; pass++) {
        for (int k = 0; k < kNumOps && r >= 0; k++) {
            if (count[k + 1] == 0 || (pass == 0) != op_is_encoder(k)) continue;
            const int64_t s = start[k];
            r = launch_op(k, lanes[op_family(k)], make_args(srcBase, gSrcOff + s, gSrcLen + s, dstBase, gDstOff + s, gDstCap + s, gOutLen + s, gStatus + s, gErr + s, (int32_t)count[k + 1]));
        }
    }
    const int32_t joined = join_lanes();
    if (r < 0) return r;
    if (joined < 0) return joined;
    HIP_TRY(achip::launch_mix_scatter(perm, nBlocks, gOutLen, gStatus, gErr, all, ctx->stream));
    return 0;
}

// ---- xxhash -------------------------------------------
int32_t achip_xxhash64_batch(achip_ctx* ctx, const void* srcBase, const int64_t* srcOff, const int32_t* srcLen, int64_t seed, int64_t* outHash, int32_t nBuffers)
{
    if (!ctx) return bad_argument("ctx is null");
    if (nBuffers < 0) return bad_argument("nBuffers < 0");
    if (nBuffers == 0) return 0;
    if (!srcOff || !srcLen || !outHash) return bad_argument("null array");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(achip::launch_xxh64_batch(srcBase, srcOff, srcLen, nBuffers, (uint64_t)seed, outHash, ctx->stream));
    return 0;
}

int32_t achip_xxhash32_batch(achip_ctx* ctx, const void* srcBase, const int64_t* srcOff, const int32_t* srcLen, int32_t seed, int32_t* outHash, int32_t nBuffers)
{
    if (!ctx) return bad_argument("ctx is null");
    if (nBuffers < 0) return bad_argument("nBuffers < 0");
    if (nBuffers == 0) return 0;
    if (!srcOff || !srcLen || !outHash) return bad_argument("null array");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(achip::launch_xxh32_batch(srcBase, srcOff, srcLen, nBuffers, (uint32_t)seed, outHash, ctx->stream));
    return 0;
}

namespace {
// one host buffer: staged to the device, hashed there, 8 bytes back
int32_t hash_host(achip_ctx* ctx, const void* src, int64_t srcLen, int64_t seed, bool wide, int64_t* out)
{
    if (!ctx) return bad_argument("ctx is null");
    if (srcLen < 0 || srcLen > 0x7FFFFFFF) return bad_argument("length out of range");
    if (srcLen > 0 && !src) return bad_argument("src is null");
    const int64_t metaOff = (srcLen + 63) & ~63LL;
    int32_t r = ensure_stage(ctx, metaOff + 64);
    if (r < 0) return r;
    uint8_t* h = ctx->hostStage;
    uint8_t* d = ctx->devStage;
    if (srcLen > 0) memcpy(h, src, (size_t)srcLen);
    *(int64_t*)(h + metaOff) = 0;                      // srcOff
    *(int32_t*)(h + metaOff + 8) = (int32_t)srcLen;    // srcLen
    *(int64_t*)(h + metaOff + 16) = 0;                 // result
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemcpyAsync(d, h, (size_t)(metaOff + 64), hipMemcpyHostToDevice, ctx->stream));
    if (wide) {
        HIP_TRY(achip::launch_xxh64_batch(d, (const int64_t*)(d + metaOff), (const int32_t*)(d + metaOff + 8), 1, (uint64_t)seed, (int64_t*)(d + metaOff + 16), ctx->stream));
    }
    else {
        HIP_TRY(achip::launch_xxh32_batch(d, (const int64_t*)(d + metaOff), (const int32_t*)(d + metaOff + 8), 1, (uint32_t)seed, (int32_t*)(d + metaOff + 16), ctx->stream));
    }
    HIP_TRY(hipMemcpyAsync(h + metaOff + 16, d + metaOff + 16, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    *out = *(int64_t*)(h + metaOff + 16);
    return 0;
}
}  // namespace

int32_t achip_xxhash64(achip_ctx* ctx, const void* src, int64_t srcLen, int64_t seed, int64_t* outHash)
{
    if (!outHash) return bad_argument("outHash is null");
    return hash_host(ctx, src, srcLen, seed, true, outHash);
}

int32_t achip_xxhash32(achip_ctx* ctx, const void* src, int64_t srcLen, int32_t seed, int32_t* outHash)
{
    if (!outHash) return bad_argument("outHash is null");
    int64_t v = 0;
    const int32_t r = hash_host(ctx, src, srcLen, seed, false, &v);
    *outHash = (int32_t)v;
    return r;
}

// ---- host-pointer batches: chunked, double-buffered staging (H2D || kernels || D2H || host copies) ---------------
// What a Compressor.compress(byte[]...) / decompress(MemorySegment...) caller gets.  The items are cut into chunks of about
// host.chunk_bytes of staging; chunk c uses slot c & 1.  Per chunk: the host gathers the inputs into the slot's pinned buffer (a few
// copy threads), `copyIn` uploads, the context stream runs the codec kernels, `copyOut` downloads, and the host scatters the outputs
// to the caller's buffers -- while the next chunk is already being gathered / uploaded / run.  Kernels stay on ONE stream (they share
// the context's scratch); events order the slots.  A mixed batch is first ordered by codec op so that every chunk is homogeneous.
struct CopyPool {
    std::vector<std::thread> threads;
    std::mutex m;
    std::condition_variable cvWork, cvDone;
    std::function<void(int64_t)> fn;
    int64_t nTasks = 0;
    std::atomic<int64_t> next{0};
    int64_t generation = 0;
    int active = 0;
    bool stop = false;

    explicit CopyPool(int n)
    {
        for (int t = 0; t < n; t++) {
            threads.emplace_back([this] { worker(); });
        }
    }
    ~CopyPool()
    {
        {
            std::lock_guard<std::mutex> g(m);
            stop = true;
        }
        cvWork.notify_all();
        for (auto& t : threads) t.join();
    }
    void worker()
    {
        int64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> g(m);
                cvWork.wait(g, [&] { return stop || generation != seen; });
                if (stop) return;
                seen = generation;
            }
            drain();
            {
                std::lock_guard<std::mutex> g(m);
                if (--active == 0) cvDone.notify_all();
            }
        }
    }
    void drain()
    {
        for (;;) {
            const int64_t i = next.fetch_add(1);
            if (i >= nTasks) return;
            fn(i);
        }
    }
    // runs f(0..n-1) on the pool's threads and the calling thread; returns when all are done
    void run(int64_t n, std::function<void(int64_t)> f)
    {
        if (n <= 0) return;
        if (threads.empty() || n == 1) {
            for (int64_t i = 0; i < n; i++) f(i);
            return;
        }
        {
            std::lock_guard<std::mutex> g(m);
            fn = std::move(f);
            nTasks = n;
            next.store(0);
            active = (int)threads.size();
            generation++;
        }
        cvWork.notify_all();
        drain();
        std::unique_lock<std::mutex> g(m);
        cvDone.wait(g, [&] { return active == 0; });
    }
};

namespace {

void destroy_host_path(achip_ctx* ctx)
{
    delete ctx->pool;
    ctx->pool = nullptr;
    delete ctx->poolOut;
    ctx->poolOut = nullptr;
    for (int s = 0; s < achip_ctx::kHostSlots; s++) {
        if (ctx->slotHost[s]) (void)hipHostFree(ctx->slotHost[s]);
        if (ctx->slotDev[s]) (void)hipFree(ctx->slotDev[s]);
        if (ctx->evH2D[s]) (void)hipEventDestroy(ctx->evH2D[s]);
        if (ctx->evK[s]) (void)hipEventDestroy(ctx->evK[s]);
        if (ctx->evD2H[s]) (void)hipEventDestroy(ctx->evD2H[s]);
    }
    if (ctx->copyIn) (void)hipStreamDestroy(ctx->copyIn);
    if (ctx->copyOut) (void)hipStreamDestroy(ctx->copyOut);
}

// `slots` staging slots of at least `slotBytes` each (pinned host + device); the copy streams, events and copy threads on first use
int32_t ensure_host_path(achip_ctx* ctx, int64_t slotBytes, int slots)
{
    HIP_TRY(hipSetDevice(ctx->device));
    if (!ctx->copyIn) {
        // The download of a chunk is a copy KERNEL of the runtime's (`__amd_rocclr_copyBuffer`: the timeline in profiles/r06_host_timeline.txt), launched wide;
        // at the decode stream's priority the next chunk's decode kernels only got the CUs when it had drained -- a chunk's kernels took 1.7 ms beside it
        // against 0.5 alone, and the pipeline ran at the sum of its stages.  The copy streams therefore have the LOWEST priority (host.copy_priority = 0: the default one).
        int least = 0, greatest = 0;
        if (ctx->hostCopyLowPriority != 0 && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest) {
            HIP_TRY(hipStreamCreateWithPriority(&ctx->copyIn, hipStreamNonBlocking, least));
            HIP_TRY(hipStreamCreateWithPriority(&ctx->copyOut, hipStreamNonBlocking, least));
        }
        else {
            (void)hipGetLastError();
            HIP_TRY(hipStreamCreateWithFlags(&ctx->copyIn, hipStreamNonBlocking));
            HIP_TRY(hipStreamCreateWithFlags(&ctx->copyOut, hipStreamNonBlocking));
        }
        for (int s = 0; s < achip_ctx::kHostSlots; s++) {
            HIP_TRY(hipEventCreateWithFlags(&ctx->evH2D[s], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&ctx->evK[s], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&ctx->evD2H[s], hipEventDisableTiming));
        }
    }
    if (!ctx->pool) {
        // gather and scatter have a pool each (they run side by side): host.copy_threads threads each, by default a sixteenth of the host's
        // hardware threads, 2 .. 16 (the calling thread / the finalizer thread is one of each pool's copiers).  (Round 6: 16 where it was 8 -- on boxes whose
        // host copies are slow, two NUMA nodes and the process on the far one, the scatter IS the call: 8 threads 31-34 GiB/s, 16 threads 33-40; on the
        // others 8 and 16 are alike: profiles/r06_notes.md)
        int t = ctx->hostCopyThreads;
        if (t == 0) t = (int)std::min<unsigned>(16u, std::max(2u, std::thread::hardware_concurrency() / 16));
        ctx->pool = new CopyPool(t - 1);
        ctx->poolOut = new CopyPool(t - 1);
    }
    if (slotBytes > ctx->slotBytes) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->copyIn));
        HIP_TRY(hipStreamSynchronize(ctx->copyOut));
        for (int s = 0; s < achip_ctx::kHostSlots; s++) {
            if (ctx->slotHost[s]) { HIP_TRY(hipHostFree(ctx->slotHost[s])); ctx->slotHost[s] = nullptr; }
            if (ctx->slotDev[s]) { HIP_TRY(hipFree(ctx->slotDev[s])); ctx->slotDev[s] = nullptr; }
        }
        ctx->slotBytes = 0;
        ctx->slotCount = 0;
        const int64_t want = std::max<int64_t>(slotBytes, 1 << 20);
        HIP_TRY(hipHostMalloc((void**)&ctx->slotHost[0], (size_t)want, hipHostMallocDefault));
        HIP_TRY(hipMalloc((void**)&ctx->slotDev[0], (size_t)want));
        ctx->slotBytes = want;
        ctx->slotCount = 1;
    }
    while (ctx->slotCount < slots) {
        const int s = ctx->slotCount;
        HIP_TRY(hipHostMalloc((void**)&ctx->slotHost[s], (size_t)ctx->slotBytes, hipHostMallocDefault));
        HIP_TRY(hipMalloc((void**)&ctx->slotDev[s], (size_t)ctx->slotBytes));
        ctx->slotCount = s + 1;
    }
    return 0;
}

// A chunk's slot: [inputs | srcOff dstOff srcLen dstCap | pad | errOffset outLen status | pad | outputs] -- ONE upload (inputs + what the kernels
// read) and ONE download (what they wrote + the outputs) per chunk.
struct HostChunk {
    int64_t first = 0, count = 0;     // range of the processing order
    int32_t op = 0;
    int64_t srcBytes = 0, dstBytes = 0;
    int64_t oSrcOff = 0, oDstOff = 0, oSrcLen = 0, oDstCap = 0, inEnd = 0;  // uploaded: [0, inEnd)
    int64_t oErr = 0, oOutLen = 0, oStatus = 0, oDst = 0, end = 0;          // downloaded: [oErr, end)
    int32_t maxLen = 0;
};

constexpr int64_t kCopyGrain = 256 << 10;  // bytes per copy task: small blocks are grouped, large ones split

// The mean output bytes per sequence over a block's first 64 sequences (LZ4 tokens: M/lz4/Lz4RawDecompressor.java:59-140; Snappy elements:
// M/snappy/SnappyRawDecompressor.java:84-110), read on the host: 1 = short (below the decoders' auto-mode thresholds: 48 bytes for LZ4, 24 for
// Snappy), 2 = long, 0 = cannot tell.  Bounds-safe on any bytes.
int probe_sequences(bool snappy, const uint8_t* p, int64_t n)
{
    int64_t at = 0, out = 0;
    int seqs = 0;
    if (snappy) {
        for (int k = 0; k < 5 && at < n; k++) {  // the uncompressed length (a varint)
            if ((p[at++] & 0x80) == 0) break;
        }
        while (seqs < 64 && at < n) {
            const int tag = p[at++];
            int64_t len;
            if ((tag & 3) == 0) {
                len = (tag >> 2) + 1;
                if (len > 60) {
                    const int extra = (int)len - 60;
                    if (at + extra > n) break;
                    len = 0;
                    for (int b = 0; b < extra; b++) len |= (int64_t)p[at + b] << (8 * b);
                    len += 1;
                    at += extra;
                }
                at += len;
            }
            else {
                len = (tag & 3) == 1 ? ((tag >> 2) & 7) + 4 : (tag >> 2) + 1;
                at += (tag & 3) == 1 ? 1 : ((tag & 3) == 2 ? 2 : 4);
            }
            out += len;
            seqs++;
        }
        return seqs < 8 ? 0 : (out < 24LL * seqs ? 1 : 2);
    }
    while (seqs < 64 && at < n) {
        const int token = p[at++];
        int64_t lit = token >> 4, ml = token & 15;
        if (lit == 15) {
            int v;
            do {
                if (at >= n) return seqs < 8 ? 0 : (out < 48LL * seqs ? 1 : 2);
                v = p[at++];
                lit += v;
            } while (v == 255);
        }
        at += lit + 2;
        if (ml == 15) {
            int v;
            do {
                if (at >= n) return seqs < 8 ? 0 : (out < 48LL * seqs ? 1 : 2);
                v = p[at++];
                ml += v;
            } while (v == 255);
        }
        out += lit + ml + 4;
        seqs++;
    }
    return seqs < 8 ? 0 : (out < 48LL * seqs ? 1 : 2);
}

// order[j] = caller's item index of the j-th processed item (nullptr: identity); ops: per item (mixed) or nullptr (all `op`)
int32_t host_batch(achip_ctx* ctx, int32_t op, const int32_t* ops, const int32_t* order, const void* srcBase, const int64_t* srcOff, const int32_t* srcLen,
                   void* dstBase, const int64_t* dstOff, const int32_t* dstCap, int32_t* outLen, int32_t* status, int64_t* errOffset, int64_t n)
{
    auto item = [&](int64_t j) -> int64_t { return order ? order[j] : j; };
    // ---- cut into chunks: homogeneous op, about hostChunkBytes of staging each, at least one item ----
    std::vector<HostChunk> chunks;
    std::vector<int64_t> sOff(n), dOff(n);  // per processed item: offsets inside its chunk's input / output regions
    int64_t maxSlot = 0;
    {
        HostChunk c;
        bool open = false;
        auto close = [&]() {
            int64_t m = (c.srcBytes + 15) & ~15LL;
            c.oSrcOff = m; m += c.count * 8;
            c.oDstOff = m; m += c.count * 8;
            c.oSrcLen = m; m += c.count * 4;
            c.oDstCap = m; m += c.count * 4;
            c.inEnd = m;
            m = (m + 63) & ~63LL;
            c.oErr = m; m += c.count * 8;
            c.oOutLen = m; m += c.count * 4;
            c.oStatus = m; m += c.count * 4;
            m = (m + 63) & ~63LL;
            c.oDst = m;
            c.end = m + c.dstBytes;
            maxSlot = std::max(maxSlot, c.end + 64);
            chunks.push_back(c);
            open = false;
        };
        // The pipeline's first stage has nothing to overlap with and neither has its last (a chunk's upload before the first kernel, the last chunk's download and
        // scatter behind everything): the first chunks are smaller -- a quarter, half of host.chunk_bytes -- and so are the last ones (half of what is left,
        // down to a quarter).  (round 6: 16 384 blocks of 64 KiB 25 -> ~21 ms per call)
        int64_t totalBytes = 0;
        for (int64_t j = 0; j < n; j++) {
            const int64_t i = item(j);
            if (srcLen[i] < 0 || dstCap[i] < 0) return bad_argument("negative length");
            totalBytes += (((int64_t)srcLen[i] + 15) & ~15LL) + (((int64_t)dstCap[i] + 15) & ~15LL);
        }
        const int64_t base = ctx->hostChunkBytes;
        const bool ramp = ctx->hostRamp != 0 && totalBytes > base;
        int64_t doneBytes = 0, limit = base;
        for (int64_t j = 0; j < n; j++) {
            const int64_t i = item(j);
            const int32_t o = ops ? ops[i] : op;
            const int64_t sb = ((int64_t)srcLen[i] + 15) & ~15LL, db = ((int64_t)dstCap[i] + 15) & ~15LL;
            if (open && (o != c.op || c.srcBytes + c.dstBytes + sb + db > limit)) close();
            if (!open) {
                c = HostChunk();
                c.first = j;
                c.op = o;
                open = true;
                if (ramp) {
                    const int64_t head = chunks.size() == 0 ? base / 4 : (chunks.size() == 1 ? base / 2 : base);
                    const int64_t tail = std::max(base / 4, (totalBytes - doneBytes) / 2);
                    limit = std::min(head, tail);
                }
            }
            doneBytes += sb + db;
            sOff[j] = c.srcBytes;
            dOff[j] = c.dstBytes;
            c.srcBytes += sb;
            c.dstBytes += db;
            c.count++;
            c.maxLen = std::max(c.maxLen, srcLen[i]);
        }
        if (open) close();
    }
    const int nSlots = (int)std::min<size_t>(chunks.size(), (size_t)ctx->hostSlots);
    int32_t r = ensure_host_path(ctx, maxSlot, nSlots);
    if (r < 0) return r;

    // copy tasks over a chunk's items: consecutive items are grouped up to kCopyGrain bytes, one task per group
    auto for_items = [&](CopyPool& pool, const HostChunk& c, bool outputs, const std::function<void(int64_t)>& body) {
        if (c.count == 1) {
            body(c.first);
            return;
        }
        std::vector<int64_t> cut;
        cut.push_back(c.first);
        int64_t acc = 0;
        for (int64_t j = c.first; j < c.first + c.count; j++) {
            const int64_t i = item(j);
            acc += outputs ? std::max(outLen[i], 0) : srcLen[i];
            if (acc >= kCopyGrain) {
                cut.push_back(j + 1);
                acc = 0;
            }
        }
        if (cut.back() != c.first + c.count) cut.push_back(c.first + c.count);
        pool.run((int64_t)cut.size() - 1, [&](int64_t t) {
            for (int64_t j = cut[t]; j < cut[t + 1]; j++) body(j);
        });
    };

    auto gather = [&](const HostChunk& c, uint8_t* h) {
        for_items(*ctx->pool, c, false, [&](int64_t j) {
            const int64_t i = item(j);
            if (srcLen[i] > 0) memcpy(h + sOff[j], (const uint8_t*)srcBase + srcOff[i], (size_t)srcLen[i]);
        });
        for (int64_t j = c.first; j < c.first + c.count; j++) {
            const int64_t i = item(j), k = j - c.first;
            ((int64_t*)(h + c.oSrcOff))[k] = sOff[j];
            ((int64_t*)(h + c.oDstOff))[k] = c.oDst + dOff[j];
            ((int32_t*)(h + c.oSrcLen))[k] = srcLen[i];
            ((int32_t*)(h + c.oDstCap))[k] = dstCap[i];
        }
    };
    auto scatter = [&](const HostChunk& c, const uint8_t* h) {
        for (int64_t j = c.first; j < c.first + c.count; j++) {
            const int64_t i = item(j), k = j - c.first;
            outLen[i] = ((const int32_t*)(h + c.oOutLen))[k];
            status[i] = ((const int32_t*)(h + c.oStatus))[k];
            if (errOffset) errOffset[i] = ((const int64_t*)(h + c.oErr))[k];
        }
        for_items(*ctx->poolOut, c, true, [&](int64_t j) {
            const int64_t i = item(j);
            if (status[i] == 0 && outLen[i] > 0) memcpy((uint8_t*)dstBase + dstOff[i], h + c.oDst + dOff[j], (size_t)outLen[i]);
        });
    };
    auto batch_args = [&](const HostChunk& c, uint8_t* d) {
        return make_args(d, (const int64_t*)(d + c.oSrcOff), (const int32_t*)(d + c.oSrcLen), d, (const int64_t*)(d + c.oDstOff), (const int32_t*)(d + c.oDstCap),
                         (int32_t*)(d + c.oOutLen), (int32_t*)(d + c.oStatus), (int64_t*)(d + c.oErr), (int32_t)c.count);
    };
    const int savedHint = ctx->maxSrcLenHint;
    // A chunk of few blocks (at most decompress.latency_max_blocks: a single block, a small batch) in host memory: a look at its first block's first tokens tells
    // the decoders apart -- short sequences: the two passes; long ones: the ring decoders' latency class.  Only the choice of the decoder depends on it, never a
    // result: whatever these bytes are, every decoder reports what the Java decoder would.  (Larger chunks -- a pipeline chunk is ~1 000 blocks -- take the two
    // passes unseen.  Looking at them too was tried: a chunk's kernels take 0.76 ms with the rings at 64 lanes against 0.92 with the two passes, but the pipeline
    // is bound by the host's copies and the link, and its rate varies 28-40 GiB/s from run to run on one box with either: nothing to gain, one more rule to explain.)
    auto look_at_tokens = [&](const HostChunk& c, const uint8_t* h) {
        const bool few = (c.op == ACHIP_OP_LZ4_DECOMPRESS || c.op == ACHIP_OP_SNAPPY_DECOMPRESS) && c.count <= std::max(ctx->latencyMaxBlocks, ctx->hostLookMaxBlocks);
        if (few) {
            ctx->smallBatchHint = probe_sequences(c.op == ACHIP_OP_SNAPPY_DECOMPRESS, h + sOff[c.first], srcLen[item(c.first)]);
        }
    };

    if (chunks.size() == 1) {
        // one chunk (a single block -- what Compressor.compress(MemorySegment, MemorySegment) hands over -- or a small batch): nothing to overlap,
        // everything in order on the context stream: upload, kernels, download, one wait
        const HostChunk& c = chunks[0];
        uint8_t* h = ctx->slotHost[0];
        uint8_t* d = ctx->slotDev[0];
        gather(c, h);
        look_at_tokens(c, h);
        HIP_TRY(hipMemcpyAsync(d, h, (size_t)c.inEnd, hipMemcpyHostToDevice, ctx->stream));
        ctx->maxSrcLenHint = std::max(c.maxLen, 1);
        r = launch_op(c.op, ctx, batch_args(c, d));
        ctx->maxSrcLenHint = savedHint;
        if (r < 0) {
            (void)hipStreamSynchronize(ctx->stream);
            return r;
        }
        HIP_TRY(hipMemcpyAsync(h + c.oErr, d + c.oErr, (size_t)(c.end - c.oErr), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        scatter(c, h);
        return 0;
    }

    // ---- several chunks: chunk c uses slot c % nSlots.  This thread gathers chunk after chunk into pinned memory and enqueues upload (copyIn),
    // kernels (the context stream: they share the context's scratch) and download (copyOut), events ordering the three; a finalizer thread waits
    // for each download and scatters the outputs to the caller's buffers with a copy pool of its own -- so that gather, upload, kernels, download
    // and scatter of up to nSlots chunks are in flight side by side. ----
    using clk = std::chrono::steady_clock;
    auto us_since = [](clk::time_point t0) { return (int64_t)std::chrono::duration_cast<std::chrono::microseconds>(clk::now() - t0).count(); };
    const clk::time_point tStart = clk::now();
    int64_t gatherUs = 0, scatterUs = 0, waitSlotUs = 0, waitDownloadUs = 0;
    std::mutex m;
    std::condition_variable cv;
    int64_t enqueued = 0, finalized = 0;  // chunk counts
    bool aborted = false;
    int32_t finalizerStatus = 0;
    std::string finalizerMessage;
    std::thread finalizer([&] {
        if (hipSetDevice(ctx->device) != hipSuccess) {
            std::lock_guard<std::mutex> g(m);
            finalizerStatus = ACHIP_STATUS(ACHIP_CLASS_DEVICE, ACHIP_D_HIP_ERROR);
            finalizerMessage = "hipSetDevice failed in the finalizer thread";
            finalized = (int64_t)chunks.size();
            cv.notify_all();
            return;
        }
        for (int64_t ci = 0; ci < (int64_t)chunks.size(); ci++) {
            {
                std::unique_lock<std::mutex> g(m);
                cv.wait(g, [&] { return enqueued > ci || aborted; });
                if (enqueued <= ci) return;
            }
            const int slot = (int)(ci % nSlots);
            clk::time_point t0 = clk::now();
            const hipError_t e = hipEventSynchronize(ctx->evD2H[slot]);
            waitDownloadUs += us_since(t0);
            if (e != hipSuccess) {
                std::lock_guard<std::mutex> g(m);
                finalizerStatus = ACHIP_STATUS(ACHIP_CLASS_DEVICE, ACHIP_D_HIP_ERROR);
                finalizerMessage = std::string("hipEventSynchronize: ") + hipGetErrorString(e);
            }
            else {
                t0 = clk::now();
                scatter(chunks[(size_t)ci], ctx->slotHost[slot]);
                scatterUs += us_since(t0);
            }
            {
                std::lock_guard<std::mutex> g(m);
                finalized = ci + 1;
            }
            cv.notify_all();
        }
    });
    auto enqueue = [&](const HostChunk& c, int slot) -> int32_t {
        uint8_t* h = ctx->slotHost[slot];
        uint8_t* d = ctx->slotDev[slot];
        if ((ctx->hostBlit & 1) != 0) HIP_TRY(achip::launch_blit(d, h, c.inEnd, ctx->hostBlitGroups, ctx->copyIn));
        else HIP_TRY(hipMemcpyAsync(d, h, (size_t)c.inEnd, hipMemcpyHostToDevice, ctx->copyIn));
        HIP_TRY(hipEventRecord(ctx->evH2D[slot], ctx->copyIn));
        HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->evH2D[slot], 0));
        ctx->maxSrcLenHint = std::max(c.maxLen, 1);
        look_at_tokens(c, h);
        const int32_t rr = launch_op(c.op, ctx, batch_args(c, d));
        ctx->maxSrcLenHint = savedHint;
        if (rr < 0) return rr;
        HIP_TRY(hipEventRecord(ctx->evK[slot], ctx->stream));
        HIP_TRY(hipStreamWaitEvent(ctx->copyOut, ctx->evK[slot], 0));
        if ((ctx->hostBlit & 2) != 0) HIP_TRY(achip::launch_blit(h + c.oErr, d + c.oErr, c.end - c.oErr, ctx->hostBlitGroups, ctx->copyOut));
        else HIP_TRY(hipMemcpyAsync(h + c.oErr, d + c.oErr, (size_t)(c.end - c.oErr), hipMemcpyDeviceToHost, ctx->copyOut));
        HIP_TRY(hipEventRecord(ctx->evD2H[slot], ctx->copyOut));
        return 0;
    };
    std::string message;
    for (int64_t ci = 0; ci < (int64_t)chunks.size() && r >= 0; ci++) {
        const int slot = (int)(ci % nSlots);
        clk::time_point t0 = clk::now();
        {
            std::unique_lock<std::mutex> g(m);
            cv.wait(g, [&] { return finalized >= ci - nSlots + 1; });  // the slot's previous chunk has left it
            if (finalizerStatus < 0) break;
        }
        waitSlotUs += us_since(t0);
        t0 = clk::now();
        gather(chunks[(size_t)ci], ctx->slotHost[slot]);
        gatherUs += us_since(t0);
        r = enqueue(chunks[(size_t)ci], slot);
        if (r < 0) message = g_lastError;
        {
            std::lock_guard<std::mutex> g(m);
            if (r < 0) aborted = true;
            else enqueued = ci + 1;
        }
        cv.notify_all();
    }
    {
        std::lock_guard<std::mutex> g(m);
        aborted = true;  // (nothing more will be enqueued: the finalizer leaves after the last enqueued chunk)
    }
    cv.notify_all();
    finalizer.join();
    ctx->hostGatherUs = gatherUs;
    ctx->hostScatterUs = scatterUs;
    ctx->hostWaitSlotUs = waitSlotUs;
    ctx->hostWaitDownloadUs = waitDownloadUs;
    ctx->hostChunks = (int64_t)chunks.size();
    ctx->hostTotalUs = us_since(tStart);
    if (r < 0 || finalizerStatus < 0) {
        (void)hipStreamSynchronize(ctx->copyIn);
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipStreamSynchronize(ctx->copyOut);
        if (r < 0) {
            g_lastError = message;
            return r;
        }
        g_lastError = finalizerMessage;
        return finalizerStatus;
    }
    // the context stream is idle again for the caller (everything it launched was awaited through evK -> evD2H)
    return 0;
}

}  // namespace

int32_t achip_batch_host(int32_t codecOp, ACHIP_BATCH_ARGS)
{
    if (!ctx) return bad_argument("ctx is null");
    if (nBlocks < 0) return bad_argument("nBlocks < 0");
    if (nBlocks == 0) return 0;
    if (codecOp < 0 || codecOp >= kNumOps) return bad_argument("unknown codecOp");
    if (!srcOff || !srcLen || !dstOff || !dstCap || !outLen || !status) return bad_argument("null metadata array");
    return host_batch(ctx, codecOp, nullptr, nullptr, srcBase, srcOff, srcLen, dstBase, dstOff, dstCap, outLen, status, errOffset, nBlocks);
}

int32_t achip_mixed_batch_host(achip_ctx* ctx, const int32_t* codecOp, const void* srcBase, const int64_t* srcOff, const int32_t* srcLen, void* dstBase,
                               const int64_t* dstOff, const int32_t* dstCap, int32_t* outLen, int32_t* status, int64_t* errOffset, int32_t nBlocks)
{
    if (!ctx) return bad_argument("ctx is null");
    if (nBlocks < 0) return bad_argument("nBlocks < 0");
    if (nBlocks == 0) return 0;
    if (!codecOp || !srcOff || !srcLen || !dstOff || !dstCap || !outLen || !status) return bad_argument("null metadata array");
    // bucket by codec op (stable): every chunk of the pipeline is then homogeneous
    std::vector<int32_t> order((size_t)nBlocks);
    int64_t start[kNumOps + 1] = {0};
    for (int32_t i = 0; i < nBlocks; i++) {
        if (codecOp[i] < 0 || codecOp[i] >= kNumOps) return bad_argument("codecOp out of range");
        start[codecOp[i] + 1]++;
    }
    for (int k = 0; k < kNumOps; k++) start[k + 1] += start[k];
    for (int32_t i = 0; i < nBlocks; i++) order[(size_t)start[codecOp[i]]++] = i;
    return host_batch(ctx, 0, codecOp, order.data(), srcBase, srcOff, srcLen, dstBase, dstOff, dstCap, outLen, status, errOffset, nBlocks);
}

// ---- single block, host pointers -----------------------------------------
static int32_t single_block(int32_t op, achip_ctx* ctx, const void* src, void* dst, int32_t srcLen, int32_t dstCap, int64_t* errOffset)
{
    if (!ctx) return bad_argument("ctx is null");
    if (srcLen < 0 || dstCap < 0) return bad_argument("negative length");
    int64_t so = 0, dofs = 0, eo = 0;
    int32_t outLen = 0, status = 0;
    int32_t r = achip_batch_host(op, ctx, src, &so, &srcLen, dst, &dofs, &dstCap, &outLen, &status, &eo, 1);
    if (r < 0) return r;
    if (errOffset) *errOffset = eo;
    return status < 0 ? status : outLen;
}

int32_t achip_snappyframed_compress(achip_ctx* ctx, const void* src, void* dst, int32_t srcLen, int32_t dstCap, int64_t* errOffset)
{
    return single_block(ACHIP_OP_SNAPPYFRAMED_COMPRESS, ctx, src, dst, srcLen, dstCap, errOffset);
}
int32_t achip_snappyframed_decompress(achip_ctx* ctx, const void* src, void* dst, int32_t srcLen, int32_t dstCap, int64_t* errOffset)
{
    return single_block(ACHIP_OP_SNAPPYFRAMED_DECOMPRESS, ctx, src, dst, srcLen, dstCap, errOffset);
}
int32_t achip_lz4hadoop_compress(achip_ctx* ctx, const void* src, void* dst, int32_t srcLen, int32_t dstCap, int64_t* errOffset)
{
    return single_block(ACHIP_OP_LZ4HADOOP_COMPRESS, ctx, src, dst, srcLen, dstCap, errOffset);
}
int32_t achip_lz4hadoop_decompress(achip_ctx* ctx, const void* src, void* dst, int32_t srcLen, int32_t dstCap, int64_t* errOffset)
{
    return single_block(ACHIP_OP_LZ4HADOOP_DECOMPRESS, ctx, src, dst, srcLen, dstCap, errOffset);
}
int32_t achip_snappyhadoop_compress(achip_ctx* ctx, const void* src, void* dst, int32_t srcLen, int32_t dstCap, int64_t* errOffset)
{
    return single_block(ACHIP_OP_SNAPPYHADOOP_COMPRESS, ctx, src, dst, srcLen, dstCap, errOffset);
}
int32_t achip_snappyhadoop_decompress(achip_ctx* ctx, const void* src, void* dst, int32_t srcLen, int32_t dstCap, int64_t* errOffset)
{
    return single_block(ACHIP_OP_SNAPPYHADOOP_DECOMPRESS, ctx, src, dst, srcLen, dstCap, errOffset);
}
int32_t achip_zstdstream_compress(achip_ctx* ctx, const void* src, void* dst, int32_t srcLen, int32_t dstCap, int64_t* errOffset)
{
    return single_block(ACHIP_OP_ZSTDSTREAM_COMPRESS, ctx, src, dst, srcLen, dstCap, errOffset);
}
int32_t achip_lz4frame_compress(achip_ctx* ctx, const void* src, void* dst, int32_t srcLen, int32_t dstCap, int64_t* errOffset)
{
    return single_block(ACHIP_OP_LZ4FRAME_COMPRESS, ctx, src, dst, srcLen, dstCap, errOffset);
}
int32_t achip_lz4frame_decompress(achip_ctx* ctx, const void* src, void* dst, int32_t srcLen, int32_t dstCap, int64_t* errOffset)
{
    return single_block(ACHIP_OP_LZ4FRAME_DECOMPRESS, ctx, src, dst, srcLen, dstCap, errOffset);
}
int32_t achip_lz4_compress(achip_ctx* ctx, const void* src, void* dst, int32_t srcLen, int32_t dstCap, int64_t* errOffset)
{
    return single_block(ACHIP_OP_LZ4_COMPRESS, ctx, src, dst, srcLen, dstCap, errOffset);
}
int32_t achip_lz4_decompress(achip_ctx* ctx, const void* src, void* dst, int32_t srcLen, int32_t dstCap, int64_t* errOffset)
{
    return single_block(ACHIP_OP_LZ4_DECOMPRESS, ctx, src, dst, srcLen, dstCap, errOffset);
}
int32_t achip_snappy_compress(achip_ctx* ctx, const void* src, void* dst, int32_t srcLen, int32_t dstCap, int64_t* errOffset)
{
    return single_block(ACHIP_OP_SNAPPY_COMPRESS, ctx, src, dst, srcLen, dstCap, errOffset);
}
int32_t achip_snappy_decompress(achip_ctx* ctx, const void* src, void* dst, int32_t srcLen, int32_t dstCap, int64_t* errOffset)
{
    return single_block(ACHIP_OP_SNAPPY_DECOMPRESS, ctx, src, dst, srcLen, dstCap, errOffset);
}
int32_t achip_zstd_compress(achip_ctx* ctx, const void* src, void* dst, int32_t srcLen, int32_t dstCap, int64_t* errOffset)
{
    return single_block(ACHIP_OP_ZSTD_COMPRESS, ctx, src, dst, srcLen, dstCap, errOffset);
}
int32_t achip_zstd_decompress(achip_ctx* ctx, const void* src, void* dst, int32_t srcLen, int32_t dstCap, int64_t* errOffset)
{
    return single_block(ACHIP_OP_ZSTD_DECOMPRESS, ctx, src, dst, srcLen, dstCap, errOffset);
}

// ---- one process, several contexts (normally one per device), one host thread each -------------------------------------------
// The native twin of java/.../HipBatchCodec.run: the batch is cut into nCtx contiguous slices balanced by srcLen + dstCap (the rule of
// achip_partition_blocks), slice d goes through achip_batch_host / achip_mixed_batch_host on ctxs[d] in a thread of its own.  Units are
// independent (M/zstd/ZstdFrameDecompressor.java:151, M/zstd/ZstdFrameCompressor.java:162): no exchange between the slices.
int32_t achip_multi_batch_host(achip_ctx* const* ctxs, int32_t nCtx, int32_t codecOp, const int32_t* codecOps, const void* srcBase, const int64_t* srcOff,
                               const int32_t* srcLen, void* dstBase, const int64_t* dstOff, const int32_t* dstCap, int32_t* outLen, int32_t* status,
                               int64_t* errOffset, int32_t nBlocks, int32_t* sliceStarts)
{
    if (!ctxs || nCtx <= 0) return bad_argument("no contexts");
    if (nCtx > 64) return bad_argument("more than 64 contexts");
    for (int32_t d = 0; d < nCtx; d++) {
        if (!ctxs[d]) return bad_argument("ctx is null");
        for (int32_t e = 0; e < d; e++) {
            if (ctxs[e] == ctxs[d]) return bad_argument("a context listed twice (a context serves one thread at a time)");
        }
    }
    if (nBlocks < 0) return bad_argument("nBlocks < 0");
    if (!codecOps && (codecOp < 0 || codecOp >= kNumOps)) return bad_argument("unknown codecOp");
    std::vector<int32_t> starts((size_t)nCtx + 1, 0);
    if (nBlocks > 0) {
        if (!srcOff || !srcLen || !dstOff || !dstCap || !outLen || !status) return bad_argument("null metadata array");
        std::vector<int64_t> weight((size_t)nBlocks);
        for (int32_t i = 0; i < nBlocks; i++) weight[(size_t)i] = (int64_t)std::max(srcLen[i], 0) + std::max(dstCap[i], 0);
        const int32_t r = achip_partition_blocks(weight.data(), nBlocks, nCtx, starts.data());
        if (r < 0) return r;
    }
    if (sliceStarts) {
        for (int32_t d = 0; d <= nCtx; d++) sliceStarts[d] = starts[(size_t)d];
    }
    std::vector<int32_t> rc((size_t)nCtx, 0);
    std::vector<std::string> messages((size_t)nCtx);
    auto slice = [&](int32_t d) {
        const int32_t first = starts[(size_t)d], count = starts[(size_t)d + 1] - first;
        if (count == 0) return;
        int64_t* eo = errOffset ? errOffset + first : nullptr;
        rc[(size_t)d] = codecOps ? achip_mixed_batch_host(ctxs[d], codecOps + first, srcBase, srcOff + first, srcLen + first, dstBase, dstOff + first, dstCap + first,
                                                          outLen + first, status + first, eo, count)
                                 : achip_batch_host(codecOp, ctxs[d], srcBase, srcOff + first, srcLen + first, dstBase, dstOff + first, dstCap + first, outLen + first,
                                                    status + first, eo, count);
        if (rc[(size_t)d] < 0) messages[(size_t)d] = g_lastError;  // (thread-local: carried to the caller's thread below)
    };
    std::vector<std::thread> workers;
    for (int32_t d = 1; d < nCtx; d++) workers.emplace_back(slice, d);
    slice(0);
    for (auto& w : workers) w.join();
    for (int32_t d = 0; d < nCtx; d++) {
        if (rc[(size_t)d] < 0) {
            g_lastError = "context " + std::to_string(d) + ": " + messages[(size_t)d];
            return rc[(size_t)d];
        }
    }
    return 0;
}

// ---- one long Zstd stream, decoded a step at a time in bounded memory ----------------------------------------------------
// What ZstdInputStream does over ZstdIncrementalFrameDecompressor (M/zstd/ZstdIncrementalFrameDecompressor.java:44-72: the states; :216-234: the
// window kept behind the output): input arrives in pieces, output leaves in pieces, a frame may be any length, frames may follow each other.
// The host walks magic / frame header / block headers (three bytes each) and hands the device STEPS of whole blocks -- up to kStepBlocks, i.e.
// 4 MiB of output -- which the pipeline's multi-block stages decode with tables, repeat offsets, window and running checksum carried from step
// to step (zstd_decompress_pipe.hip: launch_zstd_stream_step).  Memory per open stream: the window (at most kMaxWindow) twice + a step of
// output on the device, a step of input and of output on the host, the stages' scratch for one step: ~45 MB at the 8 MiB window of the Java
// writer, whatever the stream's length.
struct achip_zstd_dstream {
    static constexpr int32_t kStepBlocks = 32;
    static constexpr int32_t kStepBytes = kStepBlocks * 131072;
    static constexpr int64_t kMaxWindow = 128LL << 20;  // (window descriptors beyond 2^27: refused -- the Java frame decoder stops at 8 MiB, :303)
    bool frameDone = false;  // a frame has been read to its end: from then on the stream may end quietly between frames (the Java decoder's INITIAL state is not a stopping point)
    enum Phase { MAGIC = 0, HEADER = 1, BLOCKS = 2, FAILED = 3 };
    achip_ctx* ctx = nullptr;
    int phase = MAGIC;
    std::vector<uint8_t> pending;  // input accepted, not decoded yet
    size_t pendingAt = 0;          // ... from here on
    int64_t streamPos = 0;         // stream offset of pending[pendingAt]
    // the frame under way
    int64_t lookBack = 0;          // FrameHeader.computeRequiredOutputBufferLookBackSize
    bool hasChecksum = false;
    bool windowBeyondJava = false; // the frame's window descriptor says more than 8 MiB: its first compressed block fails as in Java (:303), RAW / RLE blocks pass
    // output decoded, not delivered yet
    uint8_t* hostOut = nullptr;    // pinned, kStepBytes
    int64_t outLen = 0, outAt = 0;
    int32_t failStatus = 0;
    int64_t failOffset = 0;
    // device
    uint8_t* hist = nullptr;       // [0, window) history (ending at `window`) | [window, window + kStepBytes) a step's output | [.., + window) room to move the history
    int64_t window = 0;            // bytes of history room
    int64_t histLen = 0;           // history bytes held (at hist + window - histLen)
    uint8_t* dSrc = nullptr;       // 6 + kStepBytes + 4 * kStepBlocks + 64
    uint8_t* hostSrc = nullptr;    // pinned, the same
    void* carry = nullptr;
    void* scratch = nullptr;
    int64_t scratchBytes = 0;
};

namespace {
constexpr int64_t kStepSrcBytes = 6 + (int64_t)achip_zstd_dstream::kStepBytes + 4 * achip_zstd_dstream::kStepBlocks + 64;

void dstream_free(achip_zstd_dstream* z)
{
    if (!z) return;
    if (z->ctx) (void)hipSetDevice(z->ctx->device);
    if (z->hostOut) (void)hipHostFree(z->hostOut);
    if (z->hostSrc) (void)hipHostFree(z->hostSrc);
    if (z->hist) (void)hipFree(z->hist);
    if (z->dSrc) (void)hipFree(z->dSrc);
    if (z->carry) (void)hipFree(z->carry);
    if (z->scratch) (void)hipFree(z->scratch);
    delete z;
}

int32_t dstream_fail(achip_zstd_dstream* z, int detail, int64_t offset)
{
    z->phase = achip_zstd_dstream::FAILED;
    z->failStatus = ACHIP_STATUS(ACHIP_CLASS_MALFORMED, detail);
    z->failOffset = offset;
    return z->failStatus;
}

// the history room for a frame whose decoder must be able to look `lookBack` bytes back (grown, never shrunk)
int32_t dstream_window(achip_zstd_dstream* z, int64_t lookBack)
{
    const int64_t want = std::max<int64_t>((lookBack + 255) & ~255LL, 1 << 16);
    if (want > z->window) {
        if (z->hist) {
            HIP_TRY(hipStreamSynchronize(z->ctx->stream));
            HIP_TRY(hipFree(z->hist));
            z->hist = nullptr;
        }
        HIP_TRY(hipMalloc((void**)&z->hist, (size_t)(2 * want + achip_zstd_dstream::kStepBytes + 256)));
        z->window = want;
    }
    z->histLen = 0;
    return 0;
}

// Decodes the blocks [first, first + blocks) that lie complete in pending (each: 3-byte header at pos[i], `stored[i]` bytes behind it).
// a block of a step: its three header bytes as the step's stand-in frame will carry them (the "last" bit is set there), its payload in `pending`, and what it
// accounts for of the STREAM's bytes (a RAW / RLE block beyond 128 KiB is handed over as several: see the BLOCKS phase of achip_zstdstream_decompress_feed)
struct StepBlock {
    int32_t header;
    size_t dataPos;
    int64_t dataLen;
    int64_t streamBytes;
};
int32_t dstream_step(achip_zstd_dstream* z, const std::vector<StepBlock>& list, bool closing, uint32_t expected)
{
    achip_ctx* ctx = z->ctx;
    const int32_t blocks = (int32_t)list.size();
    // the step as a frame of its own: magic, a descriptor that says "single segment, one byte of content size, no checksum", the size byte, the blocks
    uint8_t* h = z->hostSrc;
    const uint8_t head[6] = {0x28, 0xB5, 0x2F, 0xFD, 0x20, 0x00};
    memcpy(h, head, 6);
    int64_t at = 6;
    for (int32_t i = 0; i < blocks; i++) {
        const StepBlock& b = list[(size_t)i];
        const int32_t hd = (b.header & ~1) | (i == blocks - 1 ? 1 : 0);  // the step's last block closes the stand-in frame
        h[at] = (uint8_t)hd;
        h[at + 1] = (uint8_t)(hd >> 8);
        h[at + 2] = (uint8_t)(hd >> 16);
        memcpy(h + at + 3, z->pending.data() + b.dataPos, (size_t)b.dataLen);
        at += 3 + b.dataLen;
    }
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemcpyAsync(z->dSrc, h, (size_t)at, hipMemcpyHostToDevice, ctx->stream));
    int32_t result[3] = {0, 0, -1};
    // positions count from the oldest history byte kept: the step writes from histLen on
    uint8_t* dOut = z->hist + (z->window - z->histLen);
    HIP_TRY(achip::launch_zstd_stream_step(ctx->stream, z->scratch, z->scratchBytes, z->carry, z->dSrc, (int32_t)at, blocks, dOut, (int32_t)z->histLen,
                                           (int32_t)(z->histLen + achip_zstd_dstream::kStepBytes), closing ? 1 : 0, z->hasChecksum ? 1 : 0, expected, result));
    const int32_t good = result[0], produced = result[1];
    if (produced > 0) {
        HIP_TRY(hipMemcpyAsync(z->hostOut, z->hist + z->window, (size_t)produced, hipMemcpyDeviceToHost, ctx->stream));
        // the history for the next step: the last min(window, histLen + produced) bytes, ending where the step's output begins
        const int64_t keep = std::min<int64_t>(z->lookBack, z->histLen + produced);
        uint8_t* from = z->hist + z->window + produced - keep;
        uint8_t* to = z->hist + z->window - keep;
        if (produced >= keep) {
            HIP_TRY(hipMemcpyAsync(to, from, (size_t)keep, hipMemcpyDeviceToDevice, ctx->stream));
        }
        else {  // (the ranges overlap: by way of the room behind the step's output)
            uint8_t* tmp = z->hist + z->window + achip_zstd_dstream::kStepBytes;
            HIP_TRY(hipMemcpyAsync(tmp, from, (size_t)keep, hipMemcpyDeviceToDevice, ctx->stream));
            HIP_TRY(hipMemcpyAsync(to, tmp, (size_t)keep, hipMemcpyDeviceToDevice, ctx->stream));
        }
        z->histLen = keep;
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    z->outLen = produced;
    z->outAt = 0;
    if (good < blocks) {
        // the stream is damaged in block `good`: what lies in front of it is delivered first (the next calls), then the stream fails
        int64_t off = z->streamPos;
        for (int32_t i = 0; i < good; i++) off += list[(size_t)i].streamBytes;
        dstream_fail(z, ACHIP_D_ZSTD_CORRUPTED, off);
        return 0;
    }
    if (closing && z->hasChecksum && result[2] != 1) {
        int64_t off = z->streamPos;
        for (int32_t i = 0; i < blocks; i++) off += list[(size_t)i].streamBytes;
        dstream_fail(z, ACHIP_D_ZSTD_BAD_CHECKSUM, off + 4);  // (ZstdIncrementalFrameDecompressor.java:318-320: behind the checksum word)
        return 0;
    }
    return 0;
}
}  // namespace

void* achip_zstdstream_decompress_begin(achip_ctx* ctx)
{
    if (!ctx) {
        g_lastError = "ctx is null";
        return nullptr;
    }
    if (hipSetDevice(ctx->device) != hipSuccess) {
        g_lastError = "hipSetDevice failed";
        return nullptr;
    }
    achip_zstd_dstream* z = new achip_zstd_dstream();
    z->ctx = ctx;
    z->scratchBytes = achip::zstd_stream_step_scratch_bytes(achip_zstd_dstream::kStepBlocks);
    bool ok = hipHostMalloc((void**)&z->hostOut, (size_t)achip_zstd_dstream::kStepBytes, hipHostMallocDefault) == hipSuccess;
    ok = ok && hipHostMalloc((void**)&z->hostSrc, (size_t)kStepSrcBytes, hipHostMallocDefault) == hipSuccess;
    ok = ok && hipMalloc((void**)&z->dSrc, (size_t)kStepSrcBytes) == hipSuccess;
    ok = ok && hipMalloc(&z->carry, (size_t)achip::zstd_stream_carry_bytes()) == hipSuccess;
    ok = ok && hipMalloc(&z->scratch, (size_t)z->scratchBytes) == hipSuccess;
    if (!ok) {
        g_lastError = "out of memory for a Zstd stream's buffers";
        dstream_free(z);
        return nullptr;
    }
    return z;
}

int32_t achip_zstdstream_decompress_end(achip_ctx* ctx, void* state)
{
    (void)ctx;
    dstream_free((achip_zstd_dstream*)state);
    return 0;
}

int32_t achip_zstdstream_decompress_at_stopping_point(void* state)
{
    const achip_zstd_dstream* z = (const achip_zstd_dstream*)state;
    if (!z) return 0;
    // (state == READ_FRAME_MAGIC, whatever lies buffered: fewer than four bytes behind the last frame end the stream quietly -- ZstdInputStream.java:81-86,
    //  ZstdIncrementalFrameDecompressor.isAtStoppingPoint)
    return z->phase == achip_zstd_dstream::MAGIC && z->frameDone && z->outAt == z->outLen ? 1 : 0;
}

int32_t achip_zstdstream_decompress_feed(achip_ctx* ctx, void* state, const void* src, int64_t srcLen, void* dst, int64_t dstCap, int64_t* consumed, int64_t* produced,
                                         int64_t* errOffset)
{
    achip_zstd_dstream* z = (achip_zstd_dstream*)state;
    if (!ctx || !z || z->ctx != ctx) return bad_argument("stream state");
    if (srcLen < 0 || dstCap < 0 || (srcLen > 0 && !src) || (dstCap > 0 && !dst) || !consumed || !produced) return bad_argument("buffers");
    *consumed = 0;
    *produced = 0;
    if (errOffset) *errOffset = 0;
    const uint8_t* in = (const uint8_t*)src;
    uint8_t* out = (uint8_t*)dst;
    // what the stream may hold back of the caller's input: a step of blocks with their headers, the frame's checksum, the next frame's header
    const size_t holdLimit = (size_t)kStepSrcBytes + 64;
    for (;;) {
        // ---- decoded bytes first ----
        if (z->outAt < z->outLen) {
            const int64_t n = std::min<int64_t>(z->outLen - z->outAt, dstCap - *produced);
            if (n > 0) {
                memcpy(out + *produced, z->hostOut + z->outAt, (size_t)n);
                z->outAt += n;
                *produced += n;
            }
            if (z->outAt < z->outLen) {
                return 0;  // (the caller's buffer is full)
            }
        }
        if (z->phase == achip_zstd_dstream::FAILED) {
            if (errOffset) *errOffset = z->failOffset;
            return *produced > 0 ? 0 : z->failStatus;  // (bytes in front of the damage go out first: the NEXT call fails)
        }
        // ---- take input ----
        if (z->pendingAt > 0 && (z->pendingAt == z->pending.size() || z->pendingAt >= (size_t)(1 << 20))) {
            z->pending.erase(z->pending.begin(), z->pending.begin() + (ptrdiff_t)z->pendingAt);
            z->pendingAt = 0;
        }
        {
            const size_t held = z->pending.size() - z->pendingAt;
            const int64_t take = std::min<int64_t>(srcLen - *consumed, held < holdLimit ? (int64_t)(holdLimit - held) : 0);
            if (take > 0) {
                z->pending.insert(z->pending.end(), in + *consumed, in + *consumed + take);
                *consumed += take;
            }
        }
        const uint8_t* p = z->pending.data() + z->pendingAt;
        const int64_t have = (int64_t)(z->pending.size() - z->pendingAt);
        auto advance = [&](int64_t n) {
            z->pendingAt += (size_t)n;
            z->streamPos += n;
        };
        if (z->phase == achip_zstd_dstream::MAGIC) {
            if (have < 4) {
                return 0;  // (more input, or the end of the stream: achip_zstdstream_decompress_at_stopping_point)
            }
            const uint32_t magic = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
            if (magic != 0xFD2FB528u) {
                const int32_t st = dstream_fail(z, magic == 0xFD2FB527u ? ACHIP_D_ZSTD_V07_MAGIC : ACHIP_D_ZSTD_BAD_MAGIC, z->streamPos);
                if (errOffset) *errOffset = z->failOffset;
                return *produced > 0 ? 0 : st;
            }
            advance(4);
            z->phase = achip_zstd_dstream::HEADER;
            continue;
        }
        if (z->phase == achip_zstd_dstream::HEADER) {  // ZstdFrameDecompressor.readFrameHeader :860-947 behind determineFrameHeaderSize
            if (have < 1) return 0;
            const int32_t fhd = p[0];
            const bool singleSegment = (fhd & 0x20) != 0;
            const int32_t dictDesc = fhd & 3, csDesc = fhd >> 6;
            const int32_t headerSize = 1 + (singleSegment ? 0 : 1) + (dictDesc == 0 ? 0 : (1 << (dictDesc - 1))) + (csDesc == 0 ? (singleSegment ? 1 : 0) : (1 << csDesc));
            if (have < headerSize) return 0;
            int32_t at = 1;
            int64_t windowSize = -1;
            if (!singleSegment) {
                const int32_t wd = p[at++];
                const int64_t base = 1LL << (10 + (wd >> 3));
                windowSize = base + (base / 8) * (wd & 7);
            }
            if (dictDesc != 0) {
                const int32_t st = dstream_fail(z, ACHIP_D_ZSTD_DICTIONARY, z->streamPos + at + (1 << (dictDesc - 1)));
                if (errOffset) *errOffset = z->failOffset;
                return *produced > 0 ? 0 : st;
            }
            int64_t contentSize = -1;
            auto rd = [&](int n) {
                uint64_t v = 0;
                for (int i = 0; i < n; i++) v |= (uint64_t)p[at + i] << (8 * i);
                return v;
            };
            if (csDesc == 0) contentSize = singleSegment ? (int64_t)rd(1) : -1;
            else if (csDesc == 1) contentSize = (int64_t)rd(2) + 256;
            else if (csDesc == 2) contentSize = (int64_t)rd(4);
            else contentSize = (int64_t)rd(8);
            // FrameHeader.computeRequiredOutputBufferLookBackSize
            int64_t lookBack = contentSize < 0 ? windowSize : (windowSize < 0 ? contentSize : std::min(windowSize, contentSize));
            if (contentSize < 0 && csDesc == 3) lookBack = windowSize;  // (a content size beyond 2^63 reads negative in Java: "not set")
            // What the Java reader does with large windows, and what this one does:
            //  * a window descriptor above 8 MiB (MAX_WINDOW_SIZE): Java fails the frame's first COMPRESSED block ("Window size too large", ZstdFrameDecompressor
            //    .java:303) and copies RAW / RLE blocks as they come -- so does this reader (windowBeyondJava, checked where the blocks are listed);
            //  * a single-segment frame (no descriptor: Java's windowSize is -1 and the check above passes) of any content size: Java decodes it in a buffer
            //    that never grows beyond 8 MiB + a block (ZstdIncrementalFrameDecompressor.java:318-336) -- this reader keeps min(content size, 128 MiB) behind
            //    the position, i.e. it decodes every such stream the Java reader decodes, to the same bytes, and in addition accepts offsets between
            //    Java's buffer and 128 MiB that Java refuses;
            //  * output is handed out block by block here, where Java holds a window's worth back: an error in a later block is reported after bytes the
            //    Java reader would not have delivered yet (tests/test_gpu_zstd_stream.py: "everything the reference delivered is a prefix").
            z->windowBeyondJava = !singleSegment && windowSize > (8LL << 20);
            if (singleSegment && lookBack > achip_zstd_dstream::kMaxWindow) {
                lookBack = achip_zstd_dstream::kMaxWindow;
            }
            if (lookBack < 0 || lookBack > achip_zstd_dstream::kMaxWindow) {
                if (windowSize >= 0 && windowSize <= achip_zstd_dstream::kMaxWindow) {
                    lookBack = windowSize;
                }
                else {
                    const int32_t st = dstream_fail(z, ACHIP_D_ZSTD_WINDOW_TOO_LARGE, z->streamPos);
                    if (errOffset) *errOffset = z->failOffset;
                    return *produced > 0 ? 0 : st;
                }
            }
            z->lookBack = lookBack;
            z->hasChecksum = (fhd & 4) != 0;
            int32_t r = dstream_window(z, lookBack);
            if (r < 0) return r;
            // ZstdFrameDecompressor.reset() :199-203 and a fresh XxHash64: the carry of a new frame
            {
                std::vector<uint8_t> zero((size_t)achip::zstd_stream_carry_bytes(), 0);
                achip::zstd_stream_carry_init(zero.data());
                HIP_TRY(hipSetDevice(ctx->device));
                // (on the context's stream, where the steps run, and waited for: see achip_zstdstream_compress_begin)
                HIP_TRY(hipMemcpyAsync(z->carry, zero.data(), zero.size(), hipMemcpyHostToDevice, ctx->stream));
                HIP_TRY(hipStreamSynchronize(ctx->stream));
            }
            advance(headerSize);
            z->phase = achip_zstd_dstream::BLOCKS;
            continue;
        }
        // ---- BLOCKS: the whole blocks that lie in pending, up to a step ----
        std::vector<StepBlock> list;
        int64_t at = 0;
        bool closing = false, broken = false, brokenWindow = false;
        uint32_t expected = 0;
        while ((int32_t)list.size() < achip_zstd_dstream::kStepBlocks) {
            if (have - at < 3) break;
            const int32_t hd = p[at] | (p[at + 1] << 8) | (p[at + 2] << 16);
            const int32_t type = (hd >> 1) & 3, size = hd >> 3;
            if (type == 3 || (type == 2 && size > 131072)) {
                // ("Invalid block type" :264; a compressed block beyond Block_Maximum_Size -- "Expected match length table to be present" or worse in Java -- would outgrow a step's room)
                broken = true;
                break;
            }
            if (type == 2 && z->windowBeyondJava) {  // ("Window size too large (not yet supported)" :303: the frame's first compressed block)
                broken = true;
                brokenWindow = true;
                break;
            }
            const int64_t st = type == 1 ? 1 : size;
            if (have - at < 3 + st) break;
            const bool last = (hd & 1) != 0;
            if (last && z->hasChecksum) {
                if (have - at < 3 + st + 4) break;  // (the frame's last block goes with its checksum word)
                const uint8_t* c = p + at + 3 + st;
                expected = (uint32_t)c[0] | ((uint32_t)c[1] << 8) | ((uint32_t)c[2] << 16) | ((uint32_t)c[3] << 24);
            }
            // A RAW or RLE block may say any size up to 2^21 - 1: ZstdIncrementalFrameDecompressor.java:204-226 copies / fills that many bytes (no encoder writes such a
            // block, the format forbids it, the Java reader does not check).  Here it becomes blocks of at most 128 KiB each -- the same bytes, and such blocks touch
            // neither tables nor repeat offsets -- so that a step's room holds them; a step that has no room left for all of them ends in front of the block.
            const int32_t parts = type == 2 || size <= 131072 ? 1 : (size + 131071) / 131072;
            if ((int32_t)list.size() + parts > achip_zstd_dstream::kStepBlocks) {
                break;  // (at most 16 parts: an empty step always has room)
            }
            for (int32_t k = 0; k < parts; k++) {
                const int32_t partSize = parts == 1 ? size : std::min(131072, size - 131072 * k);
                StepBlock b;
                b.header = (hd & 6) | (partSize << 3);
                b.dataPos = z->pendingAt + (size_t)at + 3 + (type == 0 ? (size_t)131072 * (size_t)k : 0);
                b.dataLen = type == 1 ? 1 : partSize;
                b.streamBytes = (k == 0 ? 3 : 0) + (type == 1 ? (k == 0 ? 1 : 0) : partSize);
                list.push_back(b);
            }
            at += 3 + st;
            if (last) {
                closing = true;
                break;
            }
        }
        if (list.empty()) {
            if (broken) {
                const int32_t st = dstream_fail(z, brokenWindow ? ACHIP_D_ZSTD_WINDOW_TOO_LARGE : ACHIP_D_ZSTD_INVALID_BLOCK_TYPE, z->streamPos + 3);
                if (errOffset) *errOffset = z->failOffset;
                return *produced > 0 ? 0 : st;
            }
            if (*consumed == srcLen) {
                return 0;  // (a block is not whole yet: more input)
            }
            continue;  // (there was room for more of the caller's input)
        }
        const int32_t r = dstream_step(z, list, closing, expected);
        if (r < 0) return r;
        if (z->phase != achip_zstd_dstream::FAILED) {
            advance(at + (closing && z->hasChecksum ? 4 : 0));
            if (closing) {
                z->phase = achip_zstd_dstream::MAGIC;
                z->frameDone = true;
            }
        }
    }
}

// ---- ... and written a chunk at a time: ZstdOutputStream (M/zstd/ZstdOutputStream.java:93-221) in the 4 MiB it buffers --------------------------
// write() appends to the stream's buffer -- here on the device --, a full buffer is flushed (compressIfNecessary :122-131: whole blocks, the
// window and one block stay), close() writes the rest and the checksum.  One kernel step per writeChunk (zstd_stream.hip:
// zstd_ostream_step_kernel), the CompressionContext between the steps in a device-side record.  The bytes are the Java stream's whatever the
// sizes of the write() calls: the flush schedule only depends on how many bytes have arrived.
struct achip_zstd_cstream {
    static constexpr int32_t kBuffer = 4 << 20, kWindow = 1 << 20, kBlock = 131072;   // maxBufferSize = 4 x window (:52-54), level 3 / unknown size
    static constexpr int32_t kOutBytes = kBuffer + (kBuffer >> 7) + 4096;              // a step's blocks + their headers + frame header + checksum
    achip_ctx* ctx = nullptr;
    uint8_t* buf = nullptr;        // device: the stream's buffer
    uint8_t* dOut = nullptr;       // device: a step's output
    uint8_t* hostOut = nullptr;    // pinned: the same, on its way to the caller
    void* state = nullptr;
    void* slab = nullptr;
    int32_t position = 0, offset = 0;  // uncompressedPosition, uncompressedOffset
    int64_t outLen = 0, outAt = 0;
    bool finished = false;
    int32_t failStatus = 0;
};

namespace {
void cstream_free(achip_zstd_cstream* z)
{
    if (!z) return;
    if (z->ctx) (void)hipSetDevice(z->ctx->device);
    if (z->hostOut) (void)hipHostFree(z->hostOut);
    if (z->buf) (void)hipFree(z->buf);
    if (z->dOut) (void)hipFree(z->dOut);
    if (z->state) (void)hipFree(z->state);
    if (z->slab) (void)hipFree(z->slab);
    delete z;
}

// writeChunk(lastChunk) :154-221
int32_t cstream_step(achip_zstd_cstream* z, bool closing)
{
    achip_ctx* ctx = z->ctx;
    const int32_t chunk = closing ? z->position - z->offset : ((z->position - z->offset - achip_zstd_cstream::kWindow - achip_zstd_cstream::kBlock) / achip_zstd_cstream::kBlock) * achip_zstd_cstream::kBlock;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(achip::launch_zstd_ostream_step(ctx->stream, z->state, z->slab, z->buf, z->offset, chunk, closing ? 1 : 0, z->dOut, achip_zstd_cstream::kOutBytes));
    int32_t result[2] = {0, 0};  // outSize, status (ZstdOStreamState words 8 and 9)
    HIP_TRY(hipMemcpyAsync(result, (const int32_t*)z->state + 8, sizeof(result), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (result[1] != 0) {
        z->failStatus = result[1];
        return result[1];
    }
    if (result[0] > 0) {
        HIP_TRY(hipMemcpyAsync(z->hostOut, z->dOut, (size_t)result[0], hipMemcpyDeviceToHost, ctx->stream));
    }
    z->offset += chunk;
    if (!closing) {
        // the window and the bytes not yet compressed move to the buffer's front (:214-219: System.arraycopy, i.e. memmove).  Source and destination
        // OVERLAP at every flush of a full buffer (slide 1.875 MiB, 2.125 MiB to move), and an overlapping hipMemcpy is undefined: the move is made
        // in pieces of at most `slide` bytes, front to back on the one stream -- a piece's destination ends where its source begins, and a piece has
        // been read before the next one overwrites it
        const int32_t slide = z->offset - achip_zstd_cstream::kWindow;
        const int32_t toMove = achip_zstd_cstream::kWindow + (z->position - z->offset);
        for (int32_t at = 0; at < toMove && slide > 0; at += slide) {
            const int32_t n = std::min(slide, toMove - at);
            HIP_TRY(hipMemcpyAsync(z->buf + at, z->buf + at + slide, (size_t)n, hipMemcpyDeviceToDevice, ctx->stream));
        }
        z->offset -= slide;
        z->position -= slide;
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    z->outLen = result[0];
    z->outAt = 0;
    return 0;
}

int64_t cstream_deliver(achip_zstd_cstream* z, uint8_t* out, int64_t room)
{
    const int64_t n = std::min<int64_t>(z->outLen - z->outAt, room);
    if (n > 0) {
        memcpy(out, z->hostOut + z->outAt, (size_t)n);
        z->outAt += n;
    }
    return n > 0 ? n : 0;
}
}  // namespace

void* achip_zstdstream_compress_begin(achip_ctx* ctx)
{
    if (!ctx) {
        g_lastError = "ctx is null";
        return nullptr;
    }
    if (hipSetDevice(ctx->device) != hipSuccess) {
        g_lastError = "hipSetDevice failed";
        return nullptr;
    }
    achip_zstd_cstream* z = new achip_zstd_cstream();
    z->ctx = ctx;
    const size_t stateBytes = (size_t)achip::zstd_ostream_state_bytes();
    bool ok = hipHostMalloc((void**)&z->hostOut, (size_t)achip_zstd_cstream::kOutBytes, hipHostMallocDefault) == hipSuccess;
    ok = ok && hipMalloc((void**)&z->buf, (size_t)achip_zstd_cstream::kBuffer + 256) == hipSuccess;
    ok = ok && hipMalloc((void**)&z->dOut, (size_t)achip_zstd_cstream::kOutBytes) == hipSuccess;
    ok = ok && hipMalloc(&z->state, stateBytes) == hipSuccess;
    ok = ok && hipMalloc(&z->slab, (size_t)achip::zstd_ostream_slab_bytes()) == hipSuccess;
    // (on the context's stream, where the steps run: a hipMemset on the null stream is not ordered in front of work on a non-blocking stream, and an empty stream's
    // only step -- begin, close -- then met whatever the allocation held: found by tools/fuzz_zstd_stream.py as a frame without its header)
    ok = ok && hipMemsetAsync(z->state, 0, stateBytes, ctx->stream) == hipSuccess;
    if (!ok) {
        g_lastError = "out of memory for a Zstd stream's buffers";
        cstream_free(z);
        return nullptr;
    }
    return z;
}

int32_t achip_zstdstream_compress_end(achip_ctx* ctx, void* state)
{
    (void)ctx;
    cstream_free((achip_zstd_cstream*)state);
    return 0;
}

// write(src, 0, srcLen) :93-104, as far as dst has room for what the flushes on the way put out
int32_t achip_zstdstream_compress_feed(achip_ctx* ctx, void* state, const void* src, int64_t srcLen, void* dst, int64_t dstCap, int64_t* consumed, int64_t* produced)
{
    achip_zstd_cstream* z = (achip_zstd_cstream*)state;
    if (!ctx || !z || z->ctx != ctx) return bad_argument("stream state");
    if (srcLen < 0 || dstCap < 0 || (srcLen > 0 && !src) || (dstCap > 0 && !dst) || !consumed || !produced) return bad_argument("buffers");
    if (z->finished) return bad_argument("Stream is closed");
    *consumed = 0;
    *produced = 0;
    if (z->failStatus != 0) return z->failStatus;
    const uint8_t* in = (const uint8_t*)src;
    for (;;) {
        *produced += cstream_deliver(z, (uint8_t*)dst + *produced, dstCap - *produced);
        if (z->outAt < z->outLen || *consumed == srcLen) {
            return 0;  // (the caller's buffer is full, or everything is taken)
        }
        const int64_t take = std::min<int64_t>(srcLen - *consumed, achip_zstd_cstream::kBuffer - z->position);
        HIP_TRY(hipSetDevice(ctx->device));
        HIP_TRY(hipMemcpyAsync(z->buf + z->position, in + *consumed, (size_t)take, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));  // (the caller's memory is its own again when the call returns)
        z->position += (int32_t)take;
        *consumed += take;
        if (z->position == achip_zstd_cstream::kBuffer) {  // compressIfNecessary :122-131
            const int32_t r = cstream_step(z, false);
            if (r < 0) return r;
        }
    }
}

// close() :143-152: the last chunk and the checksum; returns 1 when the stream's last byte has been delivered, 0 when dst was too small for the rest
// (call again), a negative status otherwise
int32_t achip_zstdstream_compress_finish(achip_ctx* ctx, void* state, void* dst, int64_t dstCap, int64_t* produced)
{
    achip_zstd_cstream* z = (achip_zstd_cstream*)state;
    if (!ctx || !z || z->ctx != ctx) return bad_argument("stream state");
    if (dstCap < 0 || (dstCap > 0 && !dst) || !produced) return bad_argument("buffers");
    *produced = 0;
    if (z->failStatus != 0) return z->failStatus;
    *produced += cstream_deliver(z, (uint8_t*)dst, dstCap);
    if (z->outAt < z->outLen) {
        return 0;
    }
    if (!z->finished) {
        const int32_t r = cstream_step(z, true);
        if (r < 0) return r;
        z->finished = true;
        *produced += cstream_deliver(z, (uint8_t*)dst + *produced, dstCap - *produced);
    }
    return z->outAt < z->outLen ? 0 : 1;
}

// ---- multi-GPU partition (host arithmetic) --------------------------------
int32_t achip_partition_blocks(const int64_t* weight, int32_t nBlocks, int32_t nParts, int32_t* starts)
{
    if (nBlocks < 0 || nParts <= 0 || !starts) return bad_argument("bad partition arguments");
    int64_t total = 0;
    for (int32_t i = 0; i < nBlocks; i++) {
        total += weight ? std::max<int64_t>(weight[i], 0) : 1;
    }
    starts[0] = 0;
    int64_t acc = 0;
    int32_t idx = 0;
    for (int32_t p = 1; p < nParts; p++) {
        // smallest idx whose prefix weight reaches p/nParts of the total
        const __int128 target = (__int128)total * p;
        while (idx < nBlocks && (__int128)acc * nParts < target) {
            acc += weight ? std::max<int64_t>(weight[idx], 0) : 1;
            idx++;
        }
        starts[p] = idx;
    }
    starts[nParts] = nBlocks;
    return 0;
}

}  // extern "C"
