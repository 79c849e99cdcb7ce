// A look at a few words of global memory (epoch flags, counters, mask words) that stays IN FLIGHT under an MFMA stream -- without a
// destination register.
//
// Until round 5 the kernels issued such a look as `global_load_dword` inside an asm statement with a compiler-allocated output ("=v") and
// touched the value for the first time in a later asm `s_waitcnt vmcnt(0)` ("+v"): hipcc, left to itself, puts the comparison right behind
// a load it can see and waits for it there, an L2 round trip exposed in every section.  But hipcc takes an asm statement's output for valid
// the moment the statement ends.  It is free to COPY the register in front of the wait (a phi move at a branch merge: the copy then holds
// what the register held before), to RE-USE it where it can prove the value dead (the late-landing load then clobbers the new occupant),
// and a look that a short section issues but never judges stays in flight over whatever the register is given to next.  All three
// happened: lstm_upper128.hip published h_1 under the mask words of an earlier section, lstm_cluster32.hip's step 0 of
// layer 1 lost a value to a flag word -- on a process's first launch or beside a memory-bound kernel, i.e. whenever the load took longer
// than usual; every idle, warm test was exact.
//
// Here the look is ONE LDS-DMA instruction: lane i's four bytes land at LDS[zone + 4 i] (the wave's own 256-byte landing zone), nothing
// else is written.  Behind `look_landed()` the words are read back with an ordinary, compiler-visible ds_read.  There is no register for
// the compiler to copy or re-use while the load is in flight, and a look nobody judges leaves 256 stale bytes in its zone, nothing more.
// tools/check_mfma_hazards.py still scans every kernel for the old idiom (an asm load with a register destination that is touched in
// front of its wait) and fails the build on it.
#pragma once

typedef unsigned ape_desc_t __attribute__((__vector_size__(4 * sizeof(unsigned))));

// a raw buffer descriptor (scalar registers) over `bytes` bytes at `base`
__device__ __forceinline__ ape_desc_t ape_make_desc(const void* base, unsigned bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(base);
    ape_desc_t d;
    d[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
    d[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xFFFFu);
    d[2] = __builtin_amdgcn_readfirstlane(bytes);
    d[3] = 0x00020000u;
    return d;
}

// issue: lane i fetches the dword at buffer offset `voff` (its own) + `soff` (uniform) into LDS[zone_lds + 4 i]; sc1 = past the L1, like
// every load of a word another workgroup writes.  (M0 is written in the statement that reads it.  s_nop 3: the descriptor or the offset may
// have been reloaded from a spill lane by v_readlane_b32 right in front -- a VALU write of an SGPR needs five wait states before a
// vector-memory instruction reads it, and hipcc does not look inside an asm statement; tools/check_mfma_hazards.py does.)
__device__ __forceinline__ void look_issue(unsigned zone_lds, unsigned voff, ape_desc_t rsrc, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 3\n\tbuffer_load_dword %1, %2, %3 offen sc1 lds" :: "s"(zone_lds), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
// the same without sc1: words a previous KERNEL wrote
__device__ __forceinline__ void look_issue_plain(unsigned zone_lds, unsigned voff, ape_desc_t rsrc, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 3\n\tbuffer_load_dword %1, %2, %3 offen lds" :: "s"(zone_lds), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
// everything this wave has in flight has arrived -- the look in its zone, and whatever else the caller has issued
__device__ __forceinline__ void look_landed() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// The same blindness on the STORE side (round 5, lstm_upper32.hip's layer-0 form: DESIGN.md 4.18): a vector-memory store of more than 64 bits
// reads its data registers up to two wait states AFTER it issues (gfx940 and later; LLVM's GCNHazardRecognizer pads the stores hipcc emits
// itself -- "VMEM store-data hazard" -- and skips buffer stores whose soffset is an SGPR).  An asm statement is opaque to that pass: hipcc
// gave the publish store's data registers to the epoch counter (`v_add_u32 v2, 1, v51` one instruction behind `buffer_store_dwordx4 v[2:5]`),
// and beside a memory-bound neighbour sixteen lanes of a wave published the integer epoch in place of a hidden value, once in ~2000 frames.
// Every asm store of the kernels therefore carries its two wait states inside the statement (APE_STORE_TAIL); tools/check_mfma_hazards.py
// scans for a wide asm store whose data a VALU instruction writes too early and fails the build on it.
#define APE_STORE_TAIL "\n\ts_nop 1"
