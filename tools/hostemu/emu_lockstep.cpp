// tools/hostemu/emu_lockstep.cpp -- the DECODERS under the generic model of a wavefront (access-granular lockstep + earliest-in-the-program
// first: hip/hip_runtime.h) instead of the rendezvous points achip_rings.h places by hand for the emulator: a cross-check of the two --
// the ring decoders (a lane group per block, groups going their own ways inside a wavefront) must produce the same bytes either way --
// and the way to run a changed ring kernel before anybody has thought about where its lanes meet.
// STATUS (end of round 2): the lane-private decoders (ops 16, 18) and the two-pass decoders (op 24) agree with the oracle under this model as
// they do under the hand-placed points; the ring decoders with more than one lane per block (ops 44 / 46 / 48) do NOT yet (337-374 of 423
// cases differ, also with a whole wavefront per block).  Probable cause: LOOP BACK-EDGES.  "Earliest in the program first" compares
// addresses, so a lane that has already jumped back to the head of the sequence loop (low address) is preferred over lanes still in the
// previous trip's tail (high address) -- the opposite of what a wavefront does, whose lanes meet at the loop's end before any goes round
// again; the ring decoders' trips differ in length from lane to lane and hand bytes from trip to trip through the rings, the encoders' and
// two-pass decoders' loops have a cross-lane operation in every trip.  (Tried: a per-frame count of backward jumps in front of the address
// -- 337 -> 246 mismatches for op 48, so trips are part of it, but "backward" is not always "next trip" in optimised code and the encoders'
// serial-probe baselines stopped passing; taken out again.)  Until the shim knows about trips, libemu.so with achip_rings.h's own points is
// what checks the ring decoders (0 mismatches).
//   clang++ -O2 -fno-omit-frame-pointer -fsanitize-coverage=inline-8bit-counters,trace-loads,trace-stores ... -o libemu_lockstep.so emu_lockstep.cpp
//   HOSTEMU_LIB=libemu_lockstep.so python tools/hostemu/check_v3.py --ops 44,54
#define HOSTEMU_ACCESS_LOCKSTEP 1
#define HOSTEMU_NO_RINGS_LOCKSTEP 1
#include "emu.cpp"
