// tools/hostemu/emu_lockstep.cpp -- the DECODERS under the generic model of a wavefront (access-granular lockstep + earliest-in-the-program
// first: hip/hip_runtime.h) instead of the rendezvous points achip_rings.h places by hand for the emulator: a cross-check of the two --
// the ring decoders (a lane group per block, groups going their own ways inside a wavefront) must produce the same bytes either way --
// and the way to run a changed ring kernel before anybody has thought about where its lanes meet.
// RESULT (end of round 2): with NO point of achip_rings.h active the ring decoders with more than one lane per block fail (337-374 of 423
// cases): their sequence loop has no cross-lane operation per trip, lanes drift into different trips, and "earliest in the program first"
// compares addresses, not trips (a per-frame count of backward jumps helped -- 337 -> 246 -- but "backward" is not always "next trip" in
// optimised code; taken out again).  With EITHER class of points alone -- only order(), where the device has its compiler barrier
// wave_mem_order(), or only enter(), which exists for the emulator -- they pass, 0 mismatches at 4, 16 and 64 lanes per block: under
// access-granular lockstep a group only has to meet once in a while, the fine order comes from the model.  So this unit makes order()
// the group's rendezvous (HOSTEMU_RINGS_ORDER_ONLY) and nothing else: a ring kernel needs no emulator-only annotation beyond the barriers
// its device code has anyway.  Lane-private decoders (ops 16, 18) and the two-pass decoders (op 24) pass with no points at all.
#define HOSTEMU_RINGS_ORDER_ONLY 1
#define HOSTEMU_ACCESS_LOCKSTEP 1
#define HOSTEMU_NO_RINGS_LOCKSTEP 1
#include "emu.cpp"
