// tools/hostemu/emu_enc.cpp -- the ENCODERS on the CPU.  They run the same serial code on all 64 lanes of a wavefront and change shared
// state in place (hash tables, sequence stores), which is exact only under the device's lockstep: this unit is built with
//   clang++ -fno-omit-frame-pointer -fsanitize-coverage=inline-8bit-counters,trace-loads,trace-stores
// and HOSTEMU_ACCESS_LOCKSTEP, so that every memory access of the kernel source is a soft order point (hip/hip_runtime.h).  Slow (a fiber
// switch per access and lane): inputs of a few KiB take seconds, a 128 KiB Zstd block about a minute.
#define HOSTEMU_ACCESS_LOCKSTEP 1
#define HOSTEMU_ORDER_IS_RENDEZVOUS 1  // wave_mem_order(): where the source says "every lane's accesses so far come before every lane's accesses from here on" all lanes meet
#include "emu.cpp"  // (the decoders and the container readers / writers: hadoop_streams.hip, lz4_frame.hip, snappy_frame.hip)
#include "../../aircompressor_amd/csrc/lz4_compress.hip"
#include "../../aircompressor_amd/csrc/snappy_compress.hip"
#ifdef ACHIP_HOST_STATS
extern "C" { long long g_zc_stats[32]; }  // tools/hostemu/zc_stats.py
#endif
#include "../../aircompressor_amd/csrc/zstd_compress.hip"
#include "../../aircompressor_amd/csrc/zstd_stream.hip"

// op: the C ABI's operation numbers -- 1 LZ4, 3 Snappy, 5 Zstd compress; 7 LZ4 frame, 9 x-snappy-framed, 11 / 13 Hadoop LZ4 / Snappy stream writers;
// 14 the Zstd stream writer (option: 1 = chunked streams from 4 MiB on)
extern "C" int emu_encode(int op, const uint8_t* srcBase, const int64_t* srcOff, const int32_t* srcLen, uint8_t* dstBase, const int64_t* dstOff, const int32_t* dstCap,
                          int32_t* outLen, int32_t* status, int64_t* errOffset, int32_t n, int32_t option, int32_t bufferSize)
{
    achip::BatchArgs a{srcBase, srcOff, srcLen, dstBase, dstOff, dstCap, outLen, status, errOffset, n, 0};
    static std::vector<uint8_t> scratch;
    if (op == 1) {
        int maxLen = 0;
        for (int i = 0; i < n; i++) maxLen = srcLen[i] > maxLen ? srcLen[i] : maxLen;
        scratch.assign((size_t)achip::lz4_compress_scratch_bytes(), 0xCD);
        achip::g_lz4_tier_workgroups = 2;  // (the persistent grid: the emulator runs its wavefronts one after the other)
        achip::g_lz4_tier_min_blocks = 1;   // (... and its batches are small)
        achip::g_lz4_mem_waves = (option >> 4) & 3 ? ((option >> 4) & 3) - 1 : 1;  // (option bits 4, 5: 1 = one wavefront per block, 2 / 3 = one / two memory-tier wavefronts; 0 = the default)
        return achip::launch_lz4_compress(a, nullptr, option & 15, maxLen, scratch.data());
    }
    if (op == 3) {
        scratch.assign((size_t)achip::snappy_compress_scratch_bytes(n), 0xCD);
        achip::g_snappy_tier_workgroups = 2;  // (the persistent grid: the emulator runs its wavefronts one after the other)
        return achip::launch_snappy_compress(a, nullptr, option & 15, scratch.data(), (option & 16) == 0);  // (option bit 4: the sub-blocks in turn, as until round 5)
    }
    if (op == 5 || op == 14) {
        if (op == 5) a.ringPad = option == 1 ? 1 : (option == 3 ? 3 : 0);  // (what achip_abi.cpp does: the one-kernel path reads the variant from the spare field)
        scratch.assign((size_t)achip::zstd_compress_scratch_bytes(n), 0xCD);
        return op == 5 ? achip::launch_zstd_compress(a, nullptr, scratch.data(), (int64_t)scratch.size(), option) : achip::launch_zstd_stream_compress(a, nullptr, scratch.data(), option);
    }
    if (op == 7) {
        scratch.assign((size_t)achip::lz4frame_compress_scratch_bytes(n < 8 ? n : 8, false), 0xCD);  // (a slab per wavefront: the emulator runs them one after the other)
        return achip::launch_lz4frame_compress(a, nullptr, scratch.data(), (int64_t)scratch.size());
    }
    if (op == 9) {
        scratch.assign((size_t)achip::snappyframed_compress_scratch_bytes(n), 0xCD);
        return achip::launch_snappyframed_compress(a, nullptr, scratch.data(), option);
    }
    if (op == 11 || op == 13) {
        scratch.assign((size_t)achip::hadoop_compress_scratch_bytes(n), 0xCD);
        return achip::launch_hadoop_compress(a, nullptr, scratch.data(), op == 13, bufferSize);
    }
    return -1;
}

// The stream writer a chunk per launch (zstd_stream.hip: zstd_ostream_step_kernel), driven the way achip_abi.cpp's achip_zstdstream_compress_feed /
// _finish drive it: ZstdOutputStream's buffer of 4 MiB, a step per writeChunk, the buffer moved down behind a flush.  `piece`: bytes per write() call.
extern "C" int64_t emu_zstd_ostream(const uint8_t* in, int64_t n, uint8_t* out, int64_t cap, int32_t piece)
{
    constexpr int32_t kBuffer = 4 << 20, kWindow = 1 << 20, kBlock = 131072;
    std::vector<uint8_t> state((size_t)achip::zstd_ostream_state_bytes(), 0), slab((size_t)achip::zstd_ostream_slab_bytes(), 0xCD), buf((size_t)kBuffer + 64, 0xEE),
        stepOut((size_t)kBuffer + (kBuffer >> 7) + 4096);
    int32_t position = 0, offset = 0;
    int64_t produced = 0;
    auto step = [&](int32_t chunk, int32_t closing) -> int {
        achip::launch_zstd_ostream_step(nullptr, state.data(), slab.data(), buf.data(), offset, chunk, closing, stepOut.data(), (int32_t)stepOut.size());
        const int32_t* w = (const int32_t*)state.data();
        const int32_t outSize = w[8], status = w[9];
        if (status != 0) return status;
        if (produced + outSize > cap) return -2;
        memcpy(out + produced, stepOut.data(), (size_t)outSize);
        produced += outSize;
        return 0;
    };
    int64_t at = 0;
    while (at < n) {
        const int32_t take = (int32_t)std::min<int64_t>(std::min<int64_t>(n - at, piece), kBuffer - position);
        memcpy(buf.data() + position, in + at, (size_t)take);
        position += take;
        at += take;
        if (position == kBuffer) {  // compressIfNecessary :122-131
            const int32_t chunk = ((position - offset - kWindow - kBlock) / kBlock) * kBlock;
            const int r = step(chunk, 0);
            if (r != 0) return r;
            offset += chunk;
            const int32_t slide = offset - kWindow;
            memmove(buf.data(), buf.data() + slide, (size_t)(kWindow + (position - offset)));
            offset -= slide;
            position -= slide;
        }
    }
    const int r = step(position - offset, 1);
    return r != 0 ? r : produced;
}
