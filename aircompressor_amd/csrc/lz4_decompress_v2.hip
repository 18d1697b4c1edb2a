// lz4_decompress_v2.hip -- batched LZ4 block decode for gfx950, LDS-ring version (the default).
//
// Same contract and the same Java-order checks as lz4_decompress.hip (M/lz4/Lz4RawDecompressor.java:35-198);
// the difference is where the bytes live: the compressed stream is pulled from HBM once through a
// per-block LDS input ring (coalesced 16-byte granules), the output is produced into a per-block LDS
// history ring and flushed to HBM in whole aligned chunks, and back-references within LDS_REACH are
// served from LDS (achip_rings.h).  HBM traffic per block ~= compressed bytes + plaintext bytes.
#include "lz4_decode_body.h"

namespace achip {

// HANDOVER: the launch that decodes only the blocks a two-pass decode handed over (`only` filter) -- its own instantiation, so that kernel
// statistics keep it apart from the launch that decodes a whole batch
template <int GS, int IN_RING, int OUT_RING, int GPL, bool HANDOVER = false, int PHASED = 0>
__global__ __launch_bounds__(256) void lz4_decompress_rings_kernel(BatchArgs a, const int32_t* mixedGroups)
{
    // auto mode (achip_abi.cpp): both LZ4 decoders are launched, the probe's count of mixed 16-block groups picks one
    if (mixedGroups != nullptr && lz4_pick(mixedGroups, batch_count(a)) != LZ4_PICK_RINGS) {
        return;
    }
    {
        const int32_t n = batch_count(a);
        if (n < a.countLo || n >= a.countHi) {
            return;
        }
    }
    ACHIP_DYNAMIC_LDS(smem);
    constexpr int GROUPS_PER_WG = 256 / GS;
    const int g = threadIdx.x & (GS - 1);
    const int grp = threadIdx.x / GS;
    const int64_t block = (int64_t)blockIdx.x * GROUPS_PER_WG + grp;
    if (block >= batch_count(a)) {  // (a batch assembled on the device may hold fewer blocks than the launch was sized for)
        return;
    }
    if (HANDOVER) {  // the blocks a two-pass decode handed over -- if it ran at all (auto mode)
        if ((a.onlyStats != nullptr && lz4_pick(a.onlyStats, batch_count(a), a.onlyShortLimit) != LZ4_PICK_TWOPASS) || a.only[block] == 0) {
            return;
        }
    }
    const uint8_t* __restrict__ in = a.srcBase + a.srcOff[block];
    uint8_t* out = a.dstBase + a.dstOff[block];
    const int32_t inLimit = a.srcLen[block];
    const int32_t outLimit = a.dstCap[block];

    Rings<GS, IN_RING, OUT_RING, GPL, PHASED> R;
    R.init(smem + grp * (IN_RING + OUT_RING + a.ringPad), smem + grp * (IN_RING + OUT_RING + a.ringPad) + IN_RING, in, inLimit, out, g,
           a.ringPad >= 16 * GS * GPL ? smem + grp * (IN_RING + OUT_RING + a.ringPad) + IN_RING + OUT_RING : nullptr);

    int32_t st = 0;
    int32_t eo = 0;
    int32_t op = 0;
    lz4_block_decode<GS, IN_RING, OUT_RING, GPL>(R, in, inLimit, outLimit, st, eo, op);

    if (g == 0) {
        a.outLen[block] = st == 0 ? op : 0;
        a.status[block] = st;
        a.errOffset[block] = (int64_t)eo;
    }
}

template <int GS, int IN_RING, int OUT_RING, int GPL = 1, int PHASED = 0>
static hipError_t lz4d2_launch(const BatchArgs& a, hipStream_t stream, const int32_t* mixedGroups)
{
    constexpr int GROUPS_PER_WG = 256 / GS;
    const unsigned grid = (unsigned)((a.nBlocks + GROUPS_PER_WG - 1) / GROUPS_PER_WG);
    const size_t lds = (size_t)GROUPS_PER_WG * (IN_RING + OUT_RING + a.ringPad);
    if (a.only != nullptr) {
        hipLaunchKernelGGL((lz4_decompress_rings_kernel<GS, IN_RING, OUT_RING, GPL, true, PHASED>), dim3(grid), dim3(256), lds, stream, a, mixedGroups);
    }
    else {
        hipLaunchKernelGGL((lz4_decompress_rings_kernel<GS, IN_RING, OUT_RING, GPL, false, PHASED>), dim3(grid), dim3(256), lds, stream, a, mixedGroups);
    }
    return hipGetLastError();
}

// ---- FEW blocks (ring class 3, what the context picks for batches of at most `decompress.latency_max_blocks` blocks -- a single block is the
// literal Lz4HipDecompressor.decompress(MemorySegment, MemorySegment)): a workgroup of ONE wavefront per block, all 64 lanes on the block, and an
// output ring of 128 KiB -- every back-reference an LZ4 block can hold (offsets <= 65 535) is an LDS read.  With the compact rings a lone block's
// far matches (83 % of a text block's) are one memory round trip each on a wavefront that has nothing else to run: 6.3 ms for 64 KiB of text
// (profiles/r05_single_block_latency.txt); here the chain never leaves the CU.
template <int IN_RING, int OUT_RING>
__global__ __launch_bounds__(64) void lz4_decompress_latency_kernel(BatchArgs a)
{
    __shared__ __attribute__((aligned(16))) uint8_t smem[IN_RING + OUT_RING + 16];
    const int lane = threadIdx.x;
    const int64_t block = blockIdx.x;
    if (block >= batch_count(a)) {
        return;
    }
    const uint8_t* __restrict__ in = a.srcBase + a.srcOff[block];
    uint8_t* out = a.dstBase + a.dstOff[block];
    const int32_t inLimit = a.srcLen[block];
    const int32_t outLimit = a.dstCap[block];
    Rings<64, IN_RING, OUT_RING, 1> R;
    R.init(smem, smem + IN_RING, in, inLimit, out, lane, nullptr);
    int32_t st = 0;
    int32_t eo = 0;
    int32_t op = 0;
    lz4_block_decode<64, IN_RING, OUT_RING, 1>(R, in, inLimit, outLimit, st, eo, op);
    if (lane == 0) {
        a.outLen[block] = st == 0 ? op : 0;
        a.status[block] = st;
        a.errOffset[block] = (int64_t)eo;
    }
}

// ringClass: 0 = compact rings (more blocks per CU), 1 = large rings (longer LDS reach), 3 = a wavefront and 128 KiB of history per block (few blocks)
// Lanes per block for a batch of nBlocks blocks whose count the host knows: the rings want ~262 144 lanes in flight (256 CUs x 16 wavefronts), and the phased 4-lane
// form is the fastest per lane -- so 4 from 32 768 blocks on, 16 below, 64 up to 4 096 (fragments data, GiB/s at 4 / 16 / 64 lanes: 1 024 blocks of 4 MiB 50 / 71 / 85;
// 4 096 x 256 KiB 198 / 276 / 282; 8 192 x 64 KiB 386 / 519 / 298; 16 384 x 64 KiB 757 / 843 / 324; 32 768 x 64 KiB 1 301 / 912 / 350; 8 and 32 lanes never win:
// profiles/r05_groupsweep.txt)
int lz4_ring_group_for(int32_t nBlocks) { return nBlocks <= 4096 ? 64 : (nBlocks < 32768 ? 16 : 4); }

hipError_t launch_lz4_decompress_rings(const BatchArgs& a, hipStream_t stream, int groupSize, int ringClass, const int32_t* mixedGroups)
{
    if (ringClass == 3 && a.only == nullptr && mixedGroups == nullptr && a.nBlocksDev == nullptr) {
        hipLaunchKernelGGL((lz4_decompress_latency_kernel<4096, 131072>), dim3((unsigned)a.nBlocks), dim3(64), 0, stream, a);
        return hipGetLastError();
    }
    switch (groupSize) {
        case 1: return ringClass ? lz4d2_launch<1, 128, 256, 4>(a, stream, mixedGroups) : lz4d2_launch<1, 64, 128, 2>(a, stream, mixedGroups);
        case 2: return ringClass ? lz4d2_launch<2, 128, 256, 2>(a, stream, mixedGroups) : lz4d2_launch<2, 64, 128, 1>(a, stream, mixedGroups);
        case 4:
            // ring class 0 (the default): the phased form, with an input ring of four chunks (achip_rings.h); class 2: round 2's compact rings (kept for the comparison in profiles/r03_notes.md)
            switch (ringClass) {
                case 1: return lz4d2_launch<4, 256, 512>(a, stream, mixedGroups);
                case 2: return lz4d2_launch<4, 128, 256>(a, stream, mixedGroups);  // (round 2's rings: the comparison in profiles/r03_notes.md)
                // default: an input ring of four chunks, topped up once per sequence BETWEEN the literal copy and the match copy (+2.3 %: 1951 against
                // 1907 GiB/s; at the top of the loop, where Snappy's sits, it costs LZ4 6 %: profiles/r03_notes.md)
                default: return lz4d2_launch<4, 256, 256, 1, 4>(a, stream, mixedGroups);
            }
        case 8: return ringClass ? lz4d2_launch<8, 512, 1024>(a, stream, mixedGroups) : lz4d2_launch<8, 256, 512>(a, stream, mixedGroups);
        case 32: return ringClass ? lz4d2_launch<32, 2048, 4096>(a, stream, mixedGroups) : lz4d2_launch<32, 1024, 2048>(a, stream, mixedGroups);
        case 64: return ringClass ? lz4d2_launch<64, 4096, 8192>(a, stream, mixedGroups) : lz4d2_launch<64, 2048, 4096>(a, stream, mixedGroups);
        default: return ringClass ? lz4d2_launch<16, 1024, 2048>(a, stream, mixedGroups) : lz4d2_launch<16, 512, 1024>(a, stream, mixedGroups);
    }
}

}  // namespace achip
