// MLP regressor (DropoutFF, reference estimate/nn_models.py:313-370: Linear(I,256) + leaky_relu, 2 x [Linear(256,256) + leaky_relu],
// Linear(256,O)) for batches that fill the chip, eval mode: a weight-stationary TWO-STAGE PIPELINE over pairs of CUs.
//
// mlp_tile16.hip streams every layer's weights from L2 for every 16/32-row tile (52 % of the f32 MFMA peak at 262 144 rows).
// Here the weights never move: of a pair of workgroups (one per CU, the pair inside one XCD where the dispatcher allows it --
// membership by arrival ticket within the block-index class, as lstm_cluster32.hip) workgroup A keeps the input layer and the
// first hidden layer in its registers (288 per lane), workgroup B the second hidden layer and the output layer (288), and 32-row
// tiles flow A -> B through a four-slot ring in memory (32 KB per tile, in the MFMA fragment order of the consumer, copied into
// LDS by LDS-DMA).  There is no round trip anywhere -- an MLP has no recurrence -- so the hand-over latency disappears behind
// the ring's depth; each stage is 288 v_mfma_f32_32x32x2_f32 per wave and tile (two 32-unit column tiles per wave: two
// independent accumulator chains), separated by two (A) / three (B) workgroup barriers.
//
// Same arithmetic as mlp_tile16.hip up to float32 summation order.  Eval mode, last-step rows, H = 256, two hidden layers,
// I <= 32, O <= 32; everything else stays on mlp_tile16.hip.
#include "ape_internal.h"
#include "../../include/ape_hip.h"
#include <cstdlib>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((__vector_size__(4 * sizeof(unsigned))));
constexpr unsigned SPIN_LIMIT = 1u << 22;
constexpr int H = 256, KX = 32, TR = 32;            // hidden width, padded input width, rows per tile
constexpr int NSLOT = 4;                            // ring slots per pair
constexpr int HLF = (H / 8) * TR * 8;               // floats of one tile of activations [k-block 32][row 32][8] = 32 KB
constexpr int SX = KX + 4;
constexpr int CTL_FLOATS = 16 + 4 * 64;             // LDS words in front of the tiles: control words + the polls' landing zone
#ifndef APE_PIPE_KB_LOOK
#define APE_PIPE_KB_LOOK 28
#endif
#ifndef APE_PIPE_KB_FLAG
#define APE_PIPE_KB_FLAG 8
#endif
constexpr int KB_FLAG = APE_PIPE_KB_FLAG;   // k-block of layer 1 at which the flag of the tile in front goes up
// (swept at 262 144 rows, flag / look: 16 / 12 595 us, 16 / 28 591, 12 / 24 591.5, 8 / 24 589.7, 8 / 28 588.6, 4 / 28 588.6.  The consumer looks for tile i + 1's
//  flag during its tile i and the producer raises it during its tile i + 2: early flag + late look is the slack the consumer starts with)
constexpr int KB_LOOK = APE_PIPE_KB_LOOK;   // k-block of layer 2 at which the next tile's flag is looked at and its copy starts

template <bool AG>
__device__ __forceinline__ void mfma32(f32x16& acc, float w, float a) {
    if constexpr (AG) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "a"(w), "v"(a));
    else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "v"(a));
}
// The last MFMA of a stream carries its own drain (20 wait states: its 16 passes and a margin) INSIDE its asm statement.  The
// compiler knows nothing of an asm MFMA's latency and is free to put register copies of the accumulators right behind it (it
// did: a v_mov_b64 of the result two instructions after the MFMA read the lanes of the last passes too early); nothing can
// come between the instructions of one statement.  The accumulator of the other chain finished 16 passes earlier.
template <bool AG>
__device__ __forceinline__ void mfma32_last(f32x16& acc, f32x16& other, float w, float a) {
    if constexpr (AG) asm volatile("v_mfma_f32_32x32x2_f32 %0, %2, %3, %0\n\ts_nop 15\n\ts_nop 3" : "+v"(acc), "+v"(other) : "a"(w), "v"(a));
    else asm volatile("v_mfma_f32_32x32x2_f32 %0, %2, %3, %0\n\ts_nop 15\n\ts_nop 3" : "+v"(acc), "+v"(other) : "v"(w), "v"(a));
}
__device__ __forceinline__ void dma_1k(unsigned lds_addr, unsigned voff, u32x4 rsrc, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen sc1 lds"
                 :: "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

// NB k-blocks of 8 for two column tiles at once: acc0 / acc1 += W (registers w[w0 + ct * NWT + ...]) x activations (LDS, one
// ds_read_b128 per block feeds the eight MFMAs of both tiles), fragments fetched two blocks ahead
struct NoHook { __device__ __forceinline__ void operator()(int) const {} };
template <int NB, bool AG, int NW, typename Hook = NoHook>
__device__ __forceinline__ void span2(f32x16& acc0, f32x16& acc1, const float* __restrict__ src, int stride, const float (&w)[NW], int w0, int nwt,
                                      Hook hook = Hook()) {
    f32x4 a0 = *reinterpret_cast<const f32x4*>(src);
    f32x4 a1 = (NB > 1) ? *reinterpret_cast<const f32x4*>(src + stride) : a0;
    f32x4 a2 = a1;
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
        if (kb + 2 < NB) a2 = *reinterpret_cast<const f32x4*>(src + stride * (kb + 2));
        hook(kb);                                          // (memory operations of the hand-over, in the shadow of the MFMAs)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            mfma32<AG>(acc0, w[w0 + 4 * kb + j], a0[j]);
            if (kb == NB - 1 && j == 3) mfma32_last<AG>(acc1, acc0, w[w0 + nwt + 4 * kb + j], a0[j]);
            else mfma32<AG>(acc1, w[w0 + nwt + 4 * kb + j], a0[j]);
        }
        a0 = a1;
        a1 = a2;
    }
}

struct PipeParams {
    MlpParams m;
    const float* wa0;      // stage A, input layer:   [wave 4][ct 2][kb 4][lane 64][4]
    const float* wa1;      // stage A, hidden layer 1: [wave 4][ct 2][kb 32][lane 64][4]
    const float* wb2;      // stage B, hidden layer 2: same shape
    const float* wbo;      // stage B, output layer:   [wave 4][kb 8][lane 64][4]  (W_out[o = lane & 31][64 w + 8 kb + 4 hh + j], 0 for o >= O)
    float* ring;           // [pair][slot 4][HLF]
    size_t ring_bytes;
    size_t x_bytes;        // bytes of x the batch spans (< 4 GiB: one buffer descriptor)
    unsigned* ctl;         // [8 class tickets x 16][status][done][pad..][pair][full 4 x 4 | empty 4 x 4] [pair][2 XCC ids]
    unsigned diag;         // timing experiments (APE_PIPE_DIAG; results are garbage): 1 = never wait for the peer, 2 = stage A only, 4 = stage B only, 8 = segment stamps, 32 = ring stores write through
};

__global__ __launch_bounds__(256, 1) void ape_mlp_pipe(const PipeParams pp) {
    const MlpParams& p = pp.m;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, hh = lane >> 5;
    const int frag = n * 8 + hh * 4;
    const float slope = p.neg_slope;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int* ctl_s = reinterpret_cast<int*>(smem);                 // [0] abort, [1] ticket
    unsigned* look_s = reinterpret_cast<unsigned*>(smem) + 16;  // [wave 4][64]: landing zone of the polls that ride in an MFMA stream (poll_begin)
    float* lds = smem + CTL_FLOATS;

    unsigned* const class_ticket = pp.ctl;
    unsigned* const status = pp.ctl + 8 * 16;
    unsigned* const done = status + 1;
    const int cls = blockIdx.x & 7;
    if (tid == 0) {
        ctl_s[0] = 0;
        ctl_s[1] = -1;
        if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
            const unsigned tk = __hip_atomic_fetch_add(class_ticket + cls * 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (tk < gridDim.x / 8) ctl_s[1] = (int)tk;
            else __hip_atomic_store(status, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    if (ctl_s[1] < 0) return;
    const int ticket = __builtin_amdgcn_readfirstlane(ctl_s[1]);
    const int pair = (ticket >> 1) * 8 + cls, role = ticket & 1;
    const int n_pairs = gridDim.x / 2;
    // do the pair's two workgroups share an XCD (one L2)?  Block-index classes are XCDs where the dispatcher deals workgroups round
    // robin; verified here, as lstm_cluster32.hip does: each writes its XCC id, the producer reads its consumer's
    unsigned* const xcc_words = pp.ctl + 256 + (size_t)n_pairs * 32;
    unsigned my_xcc = 0u;
    if (tid == 0) {
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(my_xcc));
        my_xcc = 0x10u | (my_xcc & 0xFu);
        __hip_atomic_store(xcc_words + pair * 2 + role, my_xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // (the producer reads its consumer's id behind its prologue, when it has long been written: stage A below)
    auto same_xcd_as_peer = [&]() -> bool {
        if (tid == 0) {
            unsigned peer = my_xcc;
            if (!(pp.diag & 1u)) {
                unsigned spins = 0;
                while ((peer = __hip_atomic_load(xcc_words + pair * 2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0u) {
                    if (++spins > SPIN_LIMIT || ((spins & 255u) == 0u && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                        ctl_s[0] = 1;
                        __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            ctl_s[3] = (peer == my_xcc) ? 1 : 0;
        }
        __syncthreads();
        // ring stores: write-through (`sc1`) unless the caller opted into the plain in-XCD form AND the pair shares an XCD (ape_internal.h,
        // DESIGN.md 4.17); APE_PIPE_DIAG & 32: the write-through path whatever the flags
        const bool same = APE_HANDOVER_IN_L2(p.flags, ctl_s[3] != 0) && !(pp.diag & 32u);
        if ((pp.diag & 8u) && tid == 0 && pair < 8) pp.ctl[240 + pair * 2 + role] = 0x100u | (same ? 1u : 0u);      // (stamps: the first pairs' verdict)
        return same;
    };
    const int n_tiles = (p.N + TR - 1) / TR;
    unsigned* const full = pp.ctl + 256 + pair * 32;            // [slot][producer wave] = tiles of that slot written
    unsigned* const empty = full + 16;                          // [slot][consumer wave] = tiles of that slot copied out
    const unsigned long long ring_addr = reinterpret_cast<unsigned long long>(pp.ring);
    u32x4 ring_desc;
    ring_desc[0] = __builtin_amdgcn_readfirstlane((unsigned)ring_addr);
    ring_desc[1] = __builtin_amdgcn_readfirstlane((unsigned)(ring_addr >> 32) & 0xFFFFu);
    ring_desc[2] = __builtin_amdgcn_readfirstlane((unsigned)pp.ring_bytes);
    ring_desc[3] = 0x00020000u;
    auto slot_base = [&](int slot) -> unsigned { return (unsigned)(((size_t)pair * NSLOT + slot) * HLF * sizeof(float)); };
    auto bar = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    // all 4 words of `f` (one per peer wave) have reached `want`
    auto wait_words = [&](const unsigned* f, unsigned want) {
        if (pp.diag & 1u) return;
        unsigned spins = 0;
        while (true) {
            unsigned v = want;
            if (lane < 4) v = __hip_atomic_load(f + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__all((int)(v >= want))) return;
            if (++spins > SPIN_LIMIT || ((spins & 255u) == 0u && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                if (lane == 0) {
                    ctl_s[0] = 1;
                    __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                return;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    };
    // the same wait in two halves, so that the L2 round trip of the look runs beside the MFMAs: the four words are fetched here ... by
    // LDS-DMA into the wave's landing zone (lane i's word goes to look_s[64 wave + i]; lanes fetch word i & 3).  Round 5: NOT into a
    // compiler-allocated register of an asm statement any more -- hipcc takes such an output for valid at once and may copy it in front of
    // the wait (tools/check_mfma_hazards.py found a v_mov of this very value there; lstm_upper128.hip on what that did to its mask words)
    const unsigned long long ctl_addr = reinterpret_cast<unsigned long long>(pp.ctl);
    u32x4 ctl_desc;
    ctl_desc[0] = __builtin_amdgcn_readfirstlane((unsigned)ctl_addr);
    ctl_desc[1] = __builtin_amdgcn_readfirstlane((unsigned)(ctl_addr >> 32) & 0xFFFFu);
    ctl_desc[2] = (unsigned)((256 + (gridDim.x / 2) * 34) * sizeof(unsigned));
    ctl_desc[3] = 0x00020000u;
    const unsigned look_lds = (unsigned)reinterpret_cast<unsigned long long>(look_s) + (unsigned)(wave * 256);
    const unsigned look_voff = (unsigned)((lane & 3) * 4);
    auto poll_begin = [&](const unsigned* f) {
        const unsigned soff = __builtin_amdgcn_readfirstlane((unsigned)((f - pp.ctl) * sizeof(unsigned)));
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 3\n\tbuffer_load_dword %1, %2, %3 offen sc1 lds"
                     :: "s"(look_lds), "v"(look_voff), "s"(ctl_desc), "s"(soff) : "memory");
    };
    // ... and looked at here (behind an s_waitcnt vmcnt(0), read back from LDS); only a word that is still behind goes into the polling loop
    auto poll_landed = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
    auto poll_end = [&](const unsigned* f, unsigned want) {
        poll_landed();
        const unsigned v = look_s[wave * 64 + lane];
        if ((pp.diag & 1u) || __all((int)(v >= want))) return;
        wait_words(f, want);
    };
    // this lane's 16 values of column tile ct = units 64 wave + 32 ct + 8 (i >> 2) + 4 hh + (i & 3) of row n: four 16-byte pieces of
    // the tile layout [unit / 8][row][8]
    auto tile_off = [&](int ct, int q) -> int { return ((8 * wave + 4 * ct + q) * TR + n) * 8 + 4 * hh; };

    const unsigned lds_base = (unsigned)reinterpret_cast<unsigned long long>(lds);       // LDS byte address of `lds`
    // Everything that rides in an MFMA stream is volatile asm (or fenced), so that it stays where it is written: the compiler
    // knows neither the MFMAs' latencies nor that memory instructions are free beside them.
    // leaky_relu of a wave's 64 x 32 block in one burst BEHIND its MFMA stream: a VALU instruction between MFMAs costs 5-12
    // cycles (tools/ubench/chain32_valu.hip), about 2 on its own
    const f32x2 slope2 = {slope, slope};
    auto lrelu16 = [&](f32x16& c) {                             // (v_pk_mul_f32: two products per instruction; there is no packed f32 max)
#pragma unroll
        for (int e = 0; e < 16; e += 2) {
            const f32x2 v = {c[e], c[e + 1]};
            f32x2 t;
            float lo = c[e], hi = c[e + 1];
            asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t) : "v"(v), "v"(slope2));
            asm volatile("v_max_f32 %0, %0, %1" : "+v"(lo) : "v"(t[0]));
            asm volatile("v_max_f32 %0, %0, %1" : "+v"(hi) : "v"(t[1]));
            c[e] = lo;
            c[e + 1] = hi;
        }
    };
    auto lrelu32 = [&](f32x16& c0, f32x16& c1) {
        lrelu16(c0);
        lrelu16(c1);
    };
    // quarter q of the block into an LDS tile / a ring slot: memory instructions only, they ride in the next MFMA stream for
    // free.  lane_off = this lane's 16 bytes in k-block 8 wave of a tile; k-block 8 wave + 4 ct + q is (4 ct + q) KiB further.
    // (The ring stores name their slot in the SGPR offset: a buffer store of more than 8 bytes WITHOUT one must not be followed
    // directly by a write of its data registers -- the compiler pads its own such stores, it cannot see these.)
    const unsigned lane_off = (unsigned)(tile_off(0, 0) * 4);
#define APE_LDS_ST(OFF) asm volatile("ds_write_b128 %0, %1 offset:" #OFF :: "v"(addr), "v"(v) : "memory")
    auto lds_st = [&](unsigned addr, f32x4 v, int blk) {
        switch (blk) {
            case 0: APE_LDS_ST(0); break;
            case 1: APE_LDS_ST(1024); break;
            case 2: APE_LDS_ST(2048); break;
            case 3: APE_LDS_ST(3072); break;
            case 4: APE_LDS_ST(4096); break;
            case 5: APE_LDS_ST(5120); break;
            case 6: APE_LDS_ST(6144); break;
            default: APE_LDS_ST(7168); break;
        }
    };
#define APE_RING_ST(OFF)                                                                                                              \
    do {                                                                                                                              \
        if (wt) asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen offset:" #OFF " sc1" :: "v"(v), "v"(voff), "s"(ring_desc), "s"(base) : "memory"); \
        else asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen offset:" #OFF :: "v"(v), "v"(voff), "s"(ring_desc), "s"(base) : "memory");          \
    } while (0)
    auto ring_st = [&](unsigned voff, f32x4 v, unsigned base, int q, auto wt_tag) {
        constexpr bool wt = decltype(wt_tag)::value;
        switch (q) {
            case 0: APE_RING_ST(0); break;
            case 1: APE_RING_ST(1024); break;
            case 2: APE_RING_ST(2048); break;
            default: APE_RING_ST(3072); break;
        }
    };
    auto put_lds = [&](const f32x16& c0, const f32x16& c1, int q, unsigned tile_addr) {
        lds_st(tile_addr + lane_off, f32x4{c0[4 * q], c0[4 * q + 1], c0[4 * q + 2], c0[4 * q + 3]}, q);
        lds_st(tile_addr + lane_off, f32x4{c1[4 * q], c1[4 * q + 1], c1[4 * q + 2], c1[4 * q + 3]}, 4 + q);
    };
    const unsigned lane_off_hi = lane_off + 4096u;
    // (wt_tag: the default -- the stores write through to memory, `sc1`; false only with APE_FLAG_IN_XCD_PLAIN on a pair inside one XCD)
    auto put_ring = [&](const f32x16& c0, const f32x16& c1, int q, unsigned base, auto wt_tag) {
        ring_st(lane_off, f32x4{c0[4 * q], c0[4 * q + 1], c0[4 * q + 2], c0[4 * q + 3]}, base, q, wt_tag);
        ring_st(lane_off_hi, f32x4{c1[4 * q], c1[4 * q + 1], c1[4 * q + 2], c1[4 * q + 3]}, base, q, wt_tag);
    };
    // timing experiment (APE_PIPE_DIAG & 8): wave 0 of pair 0 sums s_memtime (shader clocks) over the segments of its loop into
    // ctl[160 + 24 role + 2 k] (tests/tools/pipe_stamps.py)
    const bool stamping = (pp.diag & 8u) != 0u && pair == 0 && wave == 0;
    unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_prev = 0;
    auto stamp = [&](int k) {
        if (!stamping) return;
        unsigned long long t;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
        if (k >= 0) seg[k] += t - t_prev;
        t_prev = t;
    };
    auto stamps_out = [&]() {
        if (stamping && lane == 0)
            for (int k = 0; k < 8; ++k) { pp.ctl[160 + 24 * role + 2 * k] = (unsigned)seg[k]; pp.ctl[160 + 24 * role + 2 * k + 1] = (unsigned)(seg[k] >> 32); }
    };
    const int my_tiles = (pair < n_tiles) ? (n_tiles - pair + n_pairs - 1) / n_pairs : 0;
    auto tile_of = [&](int j) -> int { return pair + j * n_pairs; };

    if ((pp.diag & 2u) && role == 1) {
    } else if ((pp.diag & 4u) && role == 0) {
    } else if (role == 0) {
        // =========================== stage A: x -> layer 0 -> layer 1 -> ring ==========================================
        // Software-pipelined: per iteration i ONE barrier, then layer 1 of tile i (256 MFMAs per wave) and layer 0 of tile i + 2
        // (32).  The leaky_relu + LDS commit of layer 0's result (tile i + 1) ride in the next layer-1 stream, the leaky_relu +
        // ring stores of layer 1's result in the layer-0 stream behind it: no MFMA drain, no exposed store, and every dependency
        // has a barrier and most of an iteration between its two ends.  h0 and the x tile are double-buffered by tile parity.
        float* xin = lds;                                      // [2][TR][SX]
        float* h0 = xin + 2 * TR * SX;                         // [2][HLF]
        float* bias_s = h0 + 2 * HLF;                          // [2 layers][H]
        const unsigned h0_lds = lds_base + (unsigned)(2 * TR * SX * 4);
        float w0[2 * 4 * 4];                                   // input layer: 16 registers per column tile
        float w1[2 * 32 * 4];                                  // hidden layer 1: 128 per column tile (accumulator file)
        {
            const f32x4* s0 = reinterpret_cast<const f32x4*>(pp.wa0) + (size_t)wave * 2 * 4 * 64 + lane;
#pragma unroll
            for (int i = 0; i < 8; ++i) { const f32x4 v = s0[i * 64]; w0[4 * i] = v[0]; w0[4 * i + 1] = v[1]; w0[4 * i + 2] = v[2]; w0[4 * i + 3] = v[3]; }
            const f32x4* s1 = reinterpret_cast<const f32x4*>(pp.wa1) + (size_t)wave * 2 * 32 * 64 + lane;
#pragma unroll
            for (int i = 0; i < 64; ++i) { const f32x4 v = s1[i * 64]; w1[4 * i] = v[0]; w1[4 * i + 1] = v[1]; w1[4 * i + 2] = v[2]; w1[4 * i + 3] = v[3]; }
            // every weight register "used" here: the compiler waits for the loads in front of the loop instead of carrying
            // s_waitcnt vmcnt(n) into its MFMA stream (where they would wait for this tile's ring stores as well)
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("" : "+v"(w0[i]));
#pragma unroll
            for (int i = 0; i < 256; ++i) asm volatile("" : "+a"(w1[i]));
        }
        bias_s[tid] = p.bias[0][tid];                          // (the first barrier publishes them)
        bias_s[H + tid] = p.bias[1][tid];
        auto load_bias = [&](f32x16& acc, int l, int ct) {     // unit 64 wave + 32 ct + 8 q + 4 hh + j of accumulator register 4 q + j
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_s + l * H + 64 * wave + 32 * ct + 8 * q + 4 * hh);
                acc[4 * q] = bv[0]; acc[4 * q + 1] = bv[1]; acc[4 * q + 2] = bv[2]; acc[4 * q + 3] = bv[3];
            }
        };
        // x: thread owns 4 (row, k) elements of a tile, all with the same k; f64 z-score like mlp_tile16.hip
        const bool normalize = (p.flags & APE_FLAG_NORMALIZE_INPUT) != 0;
        const int xk = tid & 31, xrow = tid >> 5;
        const double x_mean = (normalize && xk < p.I) ? p.xx_m[xk] : 0.0;
        const double x_std = (normalize && xk < p.I) ? p.xx_s[xk] : 1.0;
        float xr[4];
        // a full tile is four loads at per-lane constant offsets from a per-tile SGPR offset (no address arithmetic on the VALU; the
        // padded columns k >= I lie outside the descriptor and read as 0); only the batch's ragged last tile looks at its row numbers
        const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)pp.x_bytes, 0x00020000);
        unsigned x_voff[4];
#pragma unroll
        for (int e = 0; e < 4; ++e)
            x_voff[e] = (xk < p.I) ? (unsigned)((((size_t)(xrow + 8 * e)) * p.row_stride + p.row_offset + xk) * sizeof(float)) : 0x80000000u;
        const unsigned x_tile_bytes = (unsigned)(TR * p.row_stride * sizeof(float));
        auto fetch_x = [&](int tile) {
            const unsigned soff = (unsigned)tile * x_tile_bytes;
            if ((long long)(tile + 1) * TR <= (long long)p.N) {
#pragma unroll
                for (int e = 0; e < 4; ++e) xr[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rsrc, x_voff[e], soff, 0));
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned off = ((long long)tile * TR + xrow + 8 * e < (long long)p.N) ? x_voff[e] : 0x80000000u;
                    xr[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rsrc, off, soff, 0));
                }
            }
        };
        auto stage_x = [&](int buf) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                xin[buf * TR * SX + (xrow + 8 * e) * SX + xk] = normalize ? (float)(((double)xr[e] - x_mean) / x_std) : xr[e];
        };
        auto layer0 = [&](f32x16& c0, f32x16& c1, int buf, auto hook) {
            load_bias(c0, 0, 0);
            load_bias(c1, 0, 1);
            span2<4, false, 32>(c0, c1, xin + buf * TR * SX + n * SX + hh * 4, 8, w0, 0, 16, hook);
        };
        f32x16 a0, a1, b0, b1;                                 // layer 0's / layer 1's accumulators (two 32-unit column tiles each)
        if (my_tiles > 0) {
            // prologue: layer 0 of tiles 0 and 1 (the second one's epilogue rides in the first layer-1 stream), x of tile 2 staged
            fetch_x(tile_of(0));
            stage_x(0);
            if (my_tiles > 1) { fetch_x(tile_of(1)); stage_x(1); }
            if (my_tiles > 2) fetch_x(tile_of(2));
            bar();
            layer0(a0, a1, 0, NoHook());
            lrelu32(a0, a1);
#pragma unroll
            for (int q = 0; q < 4; ++q) put_lds(a0, a1, q, h0_lds);
            if (my_tiles > 1) {
                layer0(a0, a1, 1, NoHook());
                lrelu32(a0, a1);
            }
            bar();                                             // every wave is done with the x tiles
            if (my_tiles > 2) {
                stage_x(0);
                if (my_tiles > 3) fetch_x(tile_of(3));
            }
        }
        const bool same_xcd = same_xcd_as_peer();
        stamp(-1);
        for (int i = 0; i < my_tiles; ++i) {
            const int slot = i & (NSLOT - 1);
            bar();                                             // h0 of tile i and x of tile i + 2 complete; h0 / x buffers of the other parity free
            stamp(0);
            if (ctl_s[0] != 0) return;
            stamp(1);
            load_bias(b0, 1, 0);
            load_bias(b1, 1, 1);
            span2<32, true, 256>(b0, b1, h0 + (i & 1) * HLF + frag, TR * 8, w1, 0, 128, [&](int kb) {
                // layer 0's result of tile i + 1 into the other h0 buffer (behind the last tile: stale values nobody reads -- a branch in
                // the stream costs more than the stores)
                if (kb >= 1 && kb <= 4) put_lds(a0, a1, kb - 1, h0_lds + (unsigned)(((i + 1) & 1) * HLF * 4));
                if (kb == KB_FLAG) {
                    // the flag owed for the tile in front: its ring stores went out ~3 us ago, the x fetch behind them is as old
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (i > 1 && lane == 0)                  // (tile 0's flag went up right behind its stores, below)
                        __hip_atomic_store(full + ((i - 1) & (NSLOT - 1)) * 4 + wave, (unsigned)((i - 1) / NSLOT + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    // has the consumer copied this tile's ring slot out (tile i - NSLOT)?  Asked here, looked at behind the layer
                    poll_begin(empty + slot * 4);
                }
            });
            stamp(2);
            lrelu32(b0, b1);
            if (i >= NSLOT) poll_end(empty + slot * 4, (unsigned)(i / NSLOT));
            else poll_landed();
            // x of tile i + 3 (fetched an iteration ago) into the buffer layer 0 read an iteration ago, tile i + 4 on its way: VALU work
            // (the f64 z-score), so here between the streams; at the top of the iteration its wait would sit out the ring stores
            // just issued, here they are a whole layer 1 old
            if (i + 3 < my_tiles) {
                stage_x((i + 3) & 1);
                if (i + 4 < my_tiles) fetch_x(tile_of(i + 4));
            }
            asm volatile("" ::: "memory");
            stamp(3);
            const unsigned base = __builtin_amdgcn_readfirstlane(slot_base(slot));
            auto tail = [&](auto wt_tag) {
                if (i + 2 < my_tiles) {
                    // layer 0 of tile i + 2; layer 1's result of tile i goes into the ring beside it
                    layer0(a0, a1, i & 1, [&](int kb) {
                        if (kb == 0) { put_ring(b0, b1, 0, base, wt_tag); put_ring(b0, b1, 1, base, wt_tag); }
                        if (kb == 1) { put_ring(b0, b1, 2, base, wt_tag); put_ring(b0, b1, 3, base, wt_tag); }
                    });
                    stamp(4);
                    lrelu32(a0, a1);
                    stamp(5);
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) put_ring(b0, b1, q, base, wt_tag);
                }
            };
            if (same_xcd) tail(std::false_type{});
            else tail(std::true_type{});
            if (i == 0 && my_tiles > 1) {                      // the pair's consumer is waiting for its first tile: pay the store latency once
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) __hip_atomic_store(full + wave, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (my_tiles > 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0)
                __hip_atomic_store(full + ((my_tiles - 1) & (NSLOT - 1)) * 4 + wave, (unsigned)((my_tiles - 1) / NSLOT + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else {
        // =========================== stage B: ring -> layer 2 -> output layer -> y ======================================
        // Software-pipelined like stage A: per iteration i one barrier, layer 2 of tile i (256 MFMAs) and the output layer of
        // tile i - 1 (32, this wave's K slice); layer 2's leaky_relu + LDS commit ride in the output layer's stream, the output
        // layer's partial sums go to LDS in the next layer-2 stream and y (the sum of the four waves' partials) leaves in the
        // one after that.  h2 and the partial sums are double-buffered by tile parity.
        float* inb = lds;                                      // [2][HLF]  h1 tiles (LDS-DMA target)
        float* h2 = inb + 2 * HLF;                             // [2][HLF]
        float* bias_s = h2 + 2 * HLF;                          // [H]
        float* pbuf = bias_s + H;                              // [2][wave 4][row 32][16] partial outputs
        constexpr int PS = 16, PB = TR * PS;                   // floats per row / per wave
        const int PW = TR * p.O;                               // floats of y per tile
        const unsigned pbuf_lds = lds_base + (unsigned)((4 * HLF + H) * 4);
        const unsigned h2_lds = lds_base + (unsigned)(2 * HLF * 4);
        float w2[2 * 32 * 4];
        float wo[8 * 4];
        {
            const f32x4* s2 = reinterpret_cast<const f32x4*>(pp.wb2) + (size_t)wave * 2 * 32 * 64 + lane;
#pragma unroll
            for (int i = 0; i < 64; ++i) { const f32x4 v = s2[i * 64]; w2[4 * i] = v[0]; w2[4 * i + 1] = v[1]; w2[4 * i + 2] = v[2]; w2[4 * i + 3] = v[3]; }
            const f32x4* so = reinterpret_cast<const f32x4*>(pp.wbo) + (size_t)wave * 8 * 64 + lane;
#pragma unroll
            for (int i = 0; i < 8; ++i) { const f32x4 v = so[i * 64]; wo[4 * i] = v[0]; wo[4 * i + 1] = v[1]; wo[4 * i + 2] = v[2]; wo[4 * i + 3] = v[3]; }
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("" : "+v"(wo[i]));
#pragma unroll
            for (int i = 0; i < 256; ++i) asm volatile("" : "+a"(w2[i]));
        }
        bias_s[tid] = p.bias[2][tid];                          // (the loop's first barrier publishes them)
        float bo[16];                                          // the output bias rides in wave 0's partial sum
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int o = 8 * (i >> 2) + 4 * hh + (i & 3);
            bo[i] = (wave == 0 && o < p.O) ? p.b_out[o] : 0.0f;
        }
        auto load_bias = [&](f32x16& acc, int ct) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_s + 64 * wave + 32 * ct + 8 * q + 4 * hh);
                acc[4 * q] = bv[0]; acc[4 * q + 1] = bv[1]; acc[4 * q + 2] = bv[2]; acc[4 * q + 3] = bv[3];
            }
        };
        const unsigned inb_lds = lds_base;
        const unsigned dma_voff = (unsigned)(lane * 16);
        auto issue_copy = [&](int it_) {                       // tile it_ of this pair: ring slot -> inb[it_ & 1]; wave w copies KiB w, w + 4, ...
            const unsigned src = slot_base(it_ & (NSLOT - 1)) + (unsigned)(wave * 1024);
            const unsigned dst = inb_lds + (unsigned)((it_ & 1) * HLF * 4 + wave * 1024);
#pragma unroll
            for (int k = 0; k < 8; ++k) dma_1k(dst + (unsigned)(k * 4096), dma_voff, ring_desc, src + (unsigned)(k * 4096));
        };
        f32x16 c0, c1, acco;                                   // layer 2's accumulators; the output layer's
#pragma unroll
        for (int e = 0; e < 16; ++e) { c0[e] = 0.0f; c1[e] = 0.0f; acco[e] = 0.0f; }
        // the output layer's partial sums of tile j (held in acco) into their LDS buffer, in y order: [wave][row * O + o]
        // the output layer's partial sums of tile j (held in acco: outputs 8 q + 4 hh + j2 of row n in registers 4 q + j2) into their LDS
        // buffer: outputs 0 .. 15 as two 16-byte stores per lane
        auto write_partials = [&](int j) {
            const unsigned addr = pbuf_lds + (unsigned)((((j & 1) * 4 + wave) * PB + n * PS + 4 * hh) * 4);
            lds_st(addr, f32x4{acco[0], acco[1], acco[2], acco[3]}, 0);
            asm volatile("ds_write_b128 %0, %1 offset:32" :: "v"(addr), "v"(f32x4{acco[4], acco[5], acco[6], acco[7]}) : "memory");
        };
        // y of tile j from its four partial sums.  A tile's 32 rows of y are one contiguous run of 32 O floats; a thread owns
        // elements tid and tid + 256 of it, whose places in the partial sums and in y are per-thread constants (a full tile is LDS
        // reads, six adds and two bounds-checked stores at an SGPR offset)
        const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)((size_t)p.N * p.O * sizeof(float)), 0x00020000);
        int y_src[2];
        unsigned y_voff[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int idx = tid + 256 * k;
            y_src[k] = (idx < PW) ? (idx / p.O) * PS + (idx % p.O) : 0;
            y_voff[k] = (idx < PW) ? (unsigned)(idx * sizeof(float)) : 0x80000000u;
        }
        auto write_y = [&](int j) {
            asm volatile("" ::: "memory");
            const float* pb = pbuf + (j & 1) * 4 * PB;
            const unsigned soff = (unsigned)tile_of(j) * (unsigned)(PW * sizeof(float));
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const float v = (pb[y_src[k]] + pb[PB + y_src[k]]) + (pb[2 * PB + y_src[k]] + pb[3 * PB + y_src[k]]);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), y_rsrc, y_voff[k], soff, 0);
            }
            asm volatile("" ::: "memory");
        };
        if (my_tiles > 0) {
            wait_words(full + 0, 1u);
            issue_copy(0);
        }
        const int n_iter = (my_tiles > 0) ? my_tiles + 3 : 0;
        stamp(-1);
        for (int i = 0; i < n_iter; ++i) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's KiBs of tile i are in LDS (and older y stores are out)
            if (i < my_tiles && lane == 0)
                __hip_atomic_store(empty + (i & (NSLOT - 1)) * 4 + wave, (unsigned)(i / NSLOT + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bar();                                             // tile i in LDS; h2 of tile i - 1 and the partial sums of tile i - 3 complete
            if (ctl_s[0] != 0) return;
            stamp(0);
            if (i >= 3) write_y(i - 3);                        // (VALU work: in front of the stream, not in it)
            stamp(1);
            auto side = [&](int kb) {
                if (kb == 1) write_partials(i - 2);            // (i < 2: nothing yet, overwritten before anybody reads it)
            };
            if (i < my_tiles) {
                load_bias(c0, 0);
                load_bias(c1, 1);
                span2<32, true, 256>(c0, c1, inb + (i & 1) * HLF + frag, TR * 8, w2, 0, 128, [&](int kb) {
                    side(kb);
                    // the next tile's copy into the other buffer (its readers finished before the barrier above) starts late
                    // in this layer (KB_LOOK); the look at its flag is eight k-blocks older
                    if (kb == KB_LOOK - 8) poll_begin(full + ((i + 1) & (NSLOT - 1)) * 4);
                    if (kb == KB_LOOK && i + 1 < my_tiles) {
                        poll_end(full + ((i + 1) & (NSLOT - 1)) * 4, (unsigned)((i + 1) / NSLOT + 1));
                        issue_copy(i + 1);
                    } else if (kb == KB_LOOK) {
                        poll_landed();                                    // (the look nobody needs has landed)
                    }
                });
                stamp(2);
                lrelu32(c0, c1);
                stamp(3);
            } else if (i - 2 < my_tiles) {
                side(1);
            }
            if (i >= 1 && i - 1 < my_tiles) {
                // output layer of tile i - 1: this wave's K slice (units 64 wave .. +63) of all 32 (padded) outputs; layer 2's result
                // of tile i goes into the other h2 buffer beside it
#pragma unroll
                for (int e = 0; e < 16; ++e) acco[e] = bo[e];
                const float* src = h2 + ((i - 1) & 1) * HLF + (8 * wave) * TR * 8 + frag;
#pragma unroll
                for (int kb = 0; kb < 8; ++kb) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(src + kb * TR * 8);
                    if (kb >= 1 && kb <= 4) put_lds(c0, c1, kb - 1, h2_lds + (unsigned)((i & 1) * HLF * 4));   // (i >= my_tiles: stale values nobody reads)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (kb == 7 && j == 3) mfma32_last<false>(acco, c0, wo[4 * kb + j], a[j]);
                        else mfma32<false>(acco, wo[4 * kb + j], a[j]);
                    }
                }
                stamp(4);
            } else if (i < my_tiles) {                         // the first tile: no output layer to hide behind
#pragma unroll
                for (int q = 0; q < 4; ++q) put_lds(c0, c1, q, h2_lds);
            }
        }
    }
    stamps_out();
    // ---- self-cleaning: the last workgroup out re-zeroes every polled word ------------------------------------------
    __syncthreads();
    if (tid == 0)
        ctl_s[2] = (__hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) ? 1 : 0;
    __syncthreads();
    if (ctl_s[2] != 0) {
        for (int i = tid; i < (int)(gridDim.x / 2) * 34; i += 256) __hip_atomic_store(pp.ctl + 256 + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid < 8) __hip_atomic_store(class_ticket + tid * 16, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid == 0) __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// stage B is the larger one: 2 in-buffers + 2 h2 buffers + bias + 2 x 4 partial sums (stage A: 2 x tiles + 2 h0 buffers + 2 biases)
constexpr size_t pipe_smem(int) { return (CTL_FLOATS + (size_t)4 * HLF + H + 2 * 4 * TR * 16) * sizeof(float); }
constexpr int PIPE_MAX_O = 16;           // (two rows of y per 32 bytes ... the partial sums are kept 16 wide)

}  // namespace

size_t ape_mlp_pipe_ring_bytes(int n_cus) { return (size_t)(n_cus / 2) * NSLOT * HLF * sizeof(float); }
size_t ape_mlp_pipe_ctl_words(int n_cus) { return 256 + (size_t)(n_cus / 2) * 34; }     // + 2 XCC ids per pair
bool ape_mlp_pipe_supported(int Hd, int n_hidden, int KXd, int O) { return Hd == H && n_hidden == 2 && KXd == KX && O <= PIPE_MAX_O; }

hipError_t ape_prepare_mlp_pipe() {
    static_assert(pipe_smem(PIPE_MAX_O) <= APE_LDS_BYTES, "LDS layout exceeds a CU");
    static_assert((size_t)(CTL_FLOATS + 2 * TR * SX + 2 * HLF + 2 * H) * sizeof(float) <= pipe_smem(1), "stage A's layout exceeds stage B's");
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&ape_mlp_pipe), hipFuncAttributeMaxDynamicSharedMemorySize, APE_LDS_BYTES);
}

// one workgroup per CU, all of them resident at once (pairs hand tiles over through bounded spins): grid = the device's CU count
// rounded down to whole block-index classes of pairs
hipError_t ape_launch_mlp_pipe(const MlpParams& q, const float* wa0, const float* wa1, const float* wb2, const float* wbo, float* ring,
                               size_t ring_bytes, unsigned* ctl, int n_cus, hipStream_t stream) {
    PipeParams pp{};
#if defined(APE_CLUSTER_STAMPS) || defined(APE_ABLATE)
    // timing experiments: only the diagnostic libraries (make diag / make ablate) read the switch -- several of its bits make the
    // kernel skip waits or a whole stage (results are garbage), so the product library has no way to turn them on
    static const unsigned diag = getenv("APE_PIPE_DIAG") ? (unsigned)atoi(getenv("APE_PIPE_DIAG")) : 0u;
    pp.diag = diag;
#else
    pp.diag = 0u;
#endif
    pp.x_bytes = ((size_t)(q.N - 1) * q.row_stride + q.row_offset + q.I) * sizeof(float);
    pp.m = q; pp.wa0 = wa0; pp.wa1 = wa1; pp.wb2 = wb2; pp.wbo = wbo; pp.ring = ring; pp.ring_bytes = ring_bytes; pp.ctl = ctl;
    const int grid = (n_cus / 16) * 16;
    hipLaunchKernelGGL(ape_mlp_pipe, dim3(grid), dim3(256), pipe_smem(q.O), stream, pp);
    return hipGetLastError();
}
