// snappy_decompress_v2.hip -- batched Snappy raw-format decode for gfx950, LDS-ring version (the default).
//
// Same contract and Java-order checks as snappy_decompress.hip (M/snappy/SnappyRawDecompressor.java:35-322);
// bytes move through the per-block LDS rings of achip_rings.h (input pulled from HBM once, output
// flushed in whole aligned chunks, near back-references served from LDS).
#include "snappy_decode_body.h"

namespace achip {

// HANDOVER: the launch that decodes only the blocks a two-pass decode handed over (`only` filter) -- its own instantiation, so that kernel
// statistics keep it apart from the launch that decodes a whole batch
template <int GS, int IN_RING, int OUT_RING, int GPL, bool HANDOVER = false, int PHASED = 0>
__global__ __launch_bounds__(256) void snappy_decompress_rings_kernel(BatchArgs a, const int32_t* mixedGroups)
{
    if (mixedGroups != nullptr && snappy_pick(mixedGroups, batch_count(a)) != LZ4_PICK_RINGS) {  // auto mode (achip_abi.cpp): the lane-per-block decoder takes this batch
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
    if (HANDOVER && ((a.onlyStats != nullptr && lz4_pick(a.onlyStats, a.nBlocks, a.onlyShortLimit) != LZ4_PICK_TWOPASS) || (block < a.nBlocks && a.only[block] == 0))) {
        return;  // (the blocks a two-pass decode handed over -- if it ran at all)
    }
    if (block >= batch_count(a)) {
        return;
    }
    const uint8_t* __restrict__ in0 = a.srcBase + a.srcOff[block];
    uint8_t* out = a.dstBase + a.dstOff[block];
    const int32_t inLen0 = a.srcLen[block];
    const int32_t outLimit = a.dstCap[block];

    int32_t st = 0;
    int32_t eo = 0;
    int32_t op = 0;
    uint8_t* const slot = smem + grp * (IN_RING + OUT_RING + a.ringPad);
    snappy_buffer_decode<GS, IN_RING, OUT_RING, GPL, PHASED>(slot, slot + IN_RING, a.ringPad >= 16 * GS * GPL ? slot + IN_RING + OUT_RING : nullptr, in0, inLen0, out, outLimit, g, st,
                                                      eo, op);

    if (g == 0) {
        a.outLen[block] = st == 0 ? op : 0;
        a.status[block] = st;
        a.errOffset[block] = (int64_t)eo;
    }
}

template <int GS, int IN_RING, int OUT_RING, int GPL = 1, int PHASED = 0>
static hipError_t snd2_launch(const BatchArgs& a, hipStream_t stream, const int32_t* mixedGroups)
{
    constexpr int GROUPS_PER_WG = 256 / GS;
    const unsigned grid = (unsigned)((a.nBlocks + GROUPS_PER_WG - 1) / GROUPS_PER_WG);
    const size_t lds = (size_t)GROUPS_PER_WG * (IN_RING + OUT_RING + a.ringPad);
    if (a.only != nullptr) {
        hipLaunchKernelGGL((snappy_decompress_rings_kernel<GS, IN_RING, OUT_RING, GPL, true, PHASED>), dim3(grid), dim3(256), lds, stream, a, mixedGroups);
    }
    else {
        hipLaunchKernelGGL((snappy_decompress_rings_kernel<GS, IN_RING, OUT_RING, GPL, false, PHASED>), dim3(grid), dim3(256), lds, stream, a, mixedGroups);
    }
    return hipGetLastError();
}

// FEW blocks (ring class 3: lz4_decompress_v2.hip says why): a workgroup of one wavefront per buffer, 128 KiB of history in LDS
template <int IN_RING, int OUT_RING>
__global__ __launch_bounds__(64) void snappy_decompress_latency_kernel(BatchArgs a)
{
    __shared__ __attribute__((aligned(16))) uint8_t smem[IN_RING + OUT_RING + 16];
    const int lane = threadIdx.x;
    const int64_t block = blockIdx.x;
    if (block >= batch_count(a)) {
        return;
    }
    int32_t st = 0;
    int32_t eo = 0;
    int32_t op = 0;
    snappy_buffer_decode<64, IN_RING, OUT_RING, 1, 0>(smem, smem + IN_RING, nullptr, a.srcBase + a.srcOff[block], a.srcLen[block], a.dstBase + a.dstOff[block], a.dstCap[block], lane, st, eo, op);
    if (lane == 0) {
        a.outLen[block] = st == 0 ? op : 0;
        a.status[block] = st;
        a.errOffset[block] = (int64_t)eo;
    }
}

// Lanes per block for a batch of nBlocks blocks whose count the host knows (lz4_ring_group_for has the reasons; fragments data, GiB/s at 4 / 16 / 64 lanes:
// 1 024 blocks of 4 MiB 49 / 62 / 72; 4 096 x 256 KiB 193 / 243 / 230; 8 192 x 64 KiB 375 / 464 / 245; 16 384 x 64 KiB 740 / 746 / 268; 32 768: 1 274 / 833 / 288 --
// profiles/r05_groupsweep.txt)
int snappy_ring_group_for(int32_t nBlocks) { return nBlocks <= 2048 ? 64 : (nBlocks < 16384 ? 16 : 4); }

hipError_t launch_snappy_decompress_rings(const BatchArgs& a, hipStream_t stream, int groupSize, int ringClass, const int32_t* mixedGroups)
{
    if (ringClass == 3 && a.only == nullptr && mixedGroups == nullptr && a.nBlocksDev == nullptr) {
        hipLaunchKernelGGL((snappy_decompress_latency_kernel<4096, 131072>), dim3((unsigned)a.nBlocks), dim3(64), 0, stream, a);
        return hipGetLastError();
    }
    switch (groupSize) {
        case 1: return ringClass ? snd2_launch<1, 128, 256, 4>(a, stream, mixedGroups) : snd2_launch<1, 64, 128, 2>(a, stream, mixedGroups);
        case 2: return ringClass ? snd2_launch<2, 128, 256, 2>(a, stream, mixedGroups) : snd2_launch<2, 64, 128, 1>(a, stream, mixedGroups);
        case 4:  // ring class 0 (default): the phased form (achip_rings.h); 2: round 2's compact rings
            switch (ringClass) {
                case 1: return snd2_launch<4, 256, 512>(a, stream, mixedGroups);
                case 2: return snd2_launch<4, 128, 256>(a, stream, mixedGroups);  // (round 2's rings: the comparison in profiles/r03_notes.md)
                default: return snd2_launch<4, 256, 256, 1, 1>(a, stream, mixedGroups);  // the input ring of four chunks, topped up once per trip: +10 % (1867 against 1694 GiB/s)
            }
        case 8: return ringClass ? snd2_launch<8, 512, 1024>(a, stream, mixedGroups) : snd2_launch<8, 256, 512>(a, stream, mixedGroups);
        case 32: return ringClass ? snd2_launch<32, 2048, 4096>(a, stream, mixedGroups) : snd2_launch<32, 1024, 2048>(a, stream, mixedGroups);
        case 64: return ringClass ? snd2_launch<64, 4096, 8192>(a, stream, mixedGroups) : snd2_launch<64, 2048, 4096>(a, stream, mixedGroups);
        default: return ringClass ? snd2_launch<16, 1024, 2048>(a, stream, mixedGroups) : snd2_launch<16, 512, 1024>(a, stream, mixedGroups);
    }
}

}  // namespace achip
