// Weight-stationary kernel for the layers above layer 0 of the 3 x 128 upper-arm model in the Monte-Carlo stream bank
// (WatchPhoneUarmNN: 38 -> 3 x 128 -> 12, 50 dropout samples per frame by default; reference estimate/watch_phone_uarm_nn.py:13-41,
// estimate/nn_models.py:191-207), exact float32.  The sibling of lstm_upper32.hip (the 2 x 256 models' one layer above layer 0).
//
// nn.LSTM's dropout sits between the layers (nn_models.py:169-174): layer 0 runs once per stream (launch A of ape_streams_step), layers
// 1 and 2 over the S x n_mc sample rows.  Until round 4 they ran on the batch-tile kernel (weights re-streamed from L2 per 16-row tile:
// 62 % of the f32 MFMA peak by executed FLOP, 95 % of the bank's frame).  Here:
//
//   * TWO layers with K = 128 + 128 each are exactly the 256 accumulator-file registers per lane of a FOUR-member cluster: a member
//     (workgroup = CU) owns 32 hidden units of BOTH layers, a wave 8 of them x 4 gates = the 32 columns of one v_mfma_f32_32x32x2_f32
//     tile; the weights are the A operand and never move after the prologue.  64 clusters x 4 CUs fill the chip;
//   * persistent over 32-row tiles (cluster c owns tiles c, c + NC, ...), two tiles in flight per cluster ("sets"), and the sections
//     of a step alternate  [set 0, layer 1] [set 1, layer 1] [set 0, layer 2] [set 1, layer 2]: every hand-over -- a layer's own h for
//     its next step, layer 1's masked h for layer 2 of the same step -- has at least one whole section of the OTHER set (8.2K MFMA
//     cycles) to travel in: store -> acknowledged -> flag -> look -> LDS-DMA gather, all hung into the other set's MFMA stream;
//   * layer 1's masked input (layer 0's output under each sample's mask) arrives pre-laid in fragment order from
//     ape_mc_expand128_kernel, like lstm_upper32.hip's; that kernel also draws the keep / drop BITS of layer 1's outputs (Philox with
//     the counters of every other kernel: row quad, step, unit, layer 1).  A wave publishes its fresh h_1 twice -- plain (layer 1's
//     recurrence) and under the rows' masks (layer 2's input): one 16-byte mask word load per lane and section, four multiplies;
//   * h_{-1} = 0: step 0 of a tile is the input span alone, in both layers;
//   * the head as in lstm_upper32.hip: four more MFMAs on the fresh h_2 of the last step, partial sums per member, reduced in a fixed order.
// Exchange protocol, cluster formation (arrival tickets within the block-index class = XCD, verified at run time), bounded spins,
// sticky status word and self-cleaning are those of lstm_cluster32.hip / lstm_upper32.hip.
#include <type_traits>

#include "ape_internal.h"
#include "async_look.h"
#include "../../include/ape_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((__vector_size__(4 * sizeof(unsigned))));
constexpr unsigned SPIN_LIMIT = 1u << 22;

// v_mfma_f32_32x32x2_f32, weight operand (A) in the accumulator file (AG) or in an architectural VGPR; the accumulators are read
// only behind mfma_drain() (hipcc does not model an asm MFMA's result hazard)
template <bool AG>
__device__ __forceinline__ void mfma32(f32x16& acc, float w, float a) {
    if constexpr (AG) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "a"(w), "v"(a));
    else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "v"(a));
}
__device__ __forceinline__ void mfma_drain(f32x16& acc) { asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc)); }

// NB k-blocks of 8: acc += W (registers w[w0 ...]) x activations (LDS, one ds_read_b128 per block, `stride` floats between
// blocks), fragments fetched two blocks ahead; `mid(kb)` runs after the MFMAs of block kb (a constant after unrolling)
template <int NB, int NW, typename Mid>
__device__ __forceinline__ void span32(f32x16& acc, const float* __restrict__ src, int stride, const float (&w)[NW], int w0, Mid&& mid) {
    f32x4 a0 = *reinterpret_cast<const f32x4*>(src);
    f32x4 a1 = (NB > 1) ? *reinterpret_cast<const f32x4*>(src + stride) : a0;
    f32x4 a2 = a1;
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
        if (kb + 2 < NB) a2 = *reinterpret_cast<const f32x4*>(src + stride * (kb + 2));
#pragma unroll
        for (int j = 0; j < 4; ++j) mfma32<true>(acc, w[w0 + 4 * kb + j], a0[j]);
        mid(kb);
        a0 = a1;
        a1 = a2;
    }
}

// a flag look that does not stall the MFMA stream: LDS-DMA into the wave's landing zone, read back behind look_landed() (async_look.h)
// The mask words of a wave's eight units (32 bytes), requested early in a section and needed at its end.  They land in LDS, not in
// registers: one LDS-DMA instruction (async_look.h; lanes i and i + 8 fetch the same word), read back with an ordinary ds_read behind
// the `s_waitcnt vmcnt(0)` of the section's judge.  Round 4 had them loaded by `global_load_dwordx4` in an asm
// statement with a compiler-allocated destination ("=v") and waited for in a later asm statement -- and hipcc, which takes an asm's output
// for valid at once, put a phi COPY of those registers in front of the wait (and, on the path where it could prove the words dead, re-used
// them for an activation fragment while the load was in flight).  Whenever the load took longer than the distance to that copy -- a
// process's first launch, a memory-bound kernel on another stream -- a section published h_1 under the mask words of an EARLIER
// section: whole 32-row tiles off by 1e-4 .. 1e-2.  That was round 4's "cold-start fault" (DESIGN.md 4.17); tools/check_mfma_hazards.py
// now scans every kernel for an in-flight asm load's destination being touched, and the build fails on it.

__device__ __forceinline__ void store_16(u32x4 v, unsigned voff, u32x4 rsrc) {
    // (s_nop 4: the descriptor may have been reloaded from a spill lane by v_readlane_b32 right in front -- a VALU write of an SGPR needs
    //  five wait states before a vector-memory instruction reads it, and hipcc does not look inside an asm statement)
#if defined(UP128_PLAIN_STORES)
#if !defined(APE_UP128_ASSERT)
#error "UP128_PLAIN_STORES rebuilds round 4's faulty hand-over: for the asserting diagnostic build (tests/tools/assert_up128.py) only"
#endif
    asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %1, %2, 0 offen" APE_STORE_TAIL :: "v"(v), "v"(voff), "s"(rsrc) : "memory");
#else
    asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %1, %2, 0 offen sc1" APE_STORE_TAIL :: "v"(v), "v"(voff), "s"(rsrc) : "memory");
#endif
}
// one LDS-DMA wave-instruction: 64 lanes x 16 bytes from the buffer to 1 KiB of LDS at the wave-uniform byte address `lds_addr`
__device__ __forceinline__ void dma_1k(unsigned lds_addr, unsigned voff, u32x4 rsrc, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 3\n\tbuffer_load_dwordx4 %1, %2, %3 offen sc1 lds"        // (five wait states, as above)
                 :: "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

// hook positions of a section (k-block indices 0 .. 63; see `mid` below)
#ifndef UP128_QF
#define UP128_QF 3
#endif
#ifndef UP128_QP
#define UP128_QP 24
#endif
#ifndef UP128_QD
#define UP128_QD 4
#endif
constexpr int UH = 128;                  // hidden units of a layer = width of its input
constexpr int GH = 4;                    // members per cluster (32 units each)
constexpr int MR = 32;                   // sample rows per tile
constexpr int BH = UH / 8;               // k-blocks of 8 per span (input span and recurrent span alike)
constexpr int NWL = 4 * 2 * BH;          // weight registers per lane and layer: [W_ih | W_hh]
constexpr int NFL = 4 * GH;              // flags per (cluster, set, layer): one per member wave
constexpr int HL = GH * 4 * MR * 8;      // floats of one slice set / one input tile-step [k-block 16][row 32][8] = 16 KB
constexpr int NDMA = HL * 4 / 1024 / 4;  // LDS-DMA instructions per wave and 16 KB copy (4)
constexpr int PO = 16;                   // width of a head partial row (O <= 16)
constexpr unsigned SET_BYTES = HL * sizeof(float);

__global__ __launch_bounds__(256, 1) void ape_lstm_upper128(const Upper128Params p) {
#ifdef APE_CLUSTER_STAMPS
    const unsigned long long dg_kstart = __builtin_amdgcn_s_memtime();
#endif
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, hh = lane >> 5;     // row of the tile, half (units 4 hh .. 4 hh + 3 of the wave's 8)
    const int T = p.T, O = p.O;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xb = smem;                            // [set 2][HL]  layer 1's input of the set's current step (layer 0's output, masked), fragment order
    float* h1b = xb + 2 * HL;                    // [set 2][HL]  h_1 of the step before
    float* m1b = h1b + 2 * HL;                   // [set 2][HL]  layer 2's input: h_1 of the current step under the rows' masks
    float* h2b = m1b + 2 * HL;                   // [set 2][HL]  h_2 of the step before
    f32x4* bias_s = reinterpret_cast<f32x4*>(h2b + 2 * HL);       // [layer 2][wave 4][gate 4][hh 2]: accumulator start values (b_ih + b_hh)
    f32x4* wo_s = bias_s + 2 * 4 * 4 * 2;                         // [wave 4][lane 64]: W_out as the head MFMAs' A fragment
    float* hp = reinterpret_cast<float*>(wo_s + 4 * 64);          // [wave 4][PO][MR]: head partial sums of the four waves
    unsigned* mwl = reinterpret_cast<unsigned*>(hp + 4 * PO * MR); // [wave 4][64]: landing zones of the section's mask words ...
    unsigned* look_s = mwl + 4 * 64;                              // [wave 4][64]: ... and of its flag look (async_look.h)
    int* ctl = reinterpret_cast<int*>(look_s + 4 * 64);           // [0] abort, [1] class ticket, [2] last-out, [3] same XCD

    // control words (all zero between launches): [8 class tickets, one per 64-byte line][n_wg XCD words]
    unsigned* const class_ticket = p.xcc_slots + 64;
    unsigned* const xcc_words = p.xcc_slots + 64 + 8 * 16;
    const int cls = blockIdx.x & 7;
    unsigned my_xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(my_xcc));
    my_xcc &= 0xFu;
    if (tid == 0) {
        ctl[0] = 0;
        ctl[1] = -1;
        if (__hip_atomic_load(p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
            const unsigned tk = __hip_atomic_fetch_add(class_ticket + cls * 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (tk < gridDim.x / 8) ctl[1] = (int)tk;
            else __hip_atomic_store(p.status, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    if (ctl[1] < 0) return;
    const int ticket = __builtin_amdgcn_readfirstlane(ctl[1]);
    const int cluster = (ticket / GH) * 8 + cls, member = ticket % GH;
    const int NC = (int)gridDim.x / GH;          // clusters of this launch
    if (tid == 0)
        __hip_atomic_store(xcc_words + cluster * GH + member, 0x10u | my_xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    // ---- weights: 2 x 128 registers per lane in the accumulator file, for the whole launch.  Host layout (ape_api.hip, wup128):
    //      per layer [member 4][wave 4][register / 4][lane][4]; register 4 kb + j of lane (column m = lane & 31 = gate * 8 + unit, half hh)
    //      = [W_ih | W_hh][gate * H + member * 32 + wave * 8 + unit][8 kb + 4 hh + j]
    float w[2 * NWL];
#pragma unroll
    for (int l = 0; l < 2; ++l) {
        const f32x4* s1 = reinterpret_cast<const f32x4*>(p.w[l]) + ((size_t)(member * 4 + wave) * (NWL / 4)) * 64 + lane;
#pragma unroll
        for (int i = 0; i < NWL / 4; ++i) {
            const f32x4 v = s1[i * 64];
            w[l * NWL + 4 * i] = v[0]; w[l * NWL + 4 * i + 1] = v[1]; w[l * NWL + 4 * i + 2] = v[2]; w[l * NWL + 4 * i + 3] = v[3];
        }
    }
    // accumulator start values: registers 4 gate + j <-> unit member*32 + wave*8 + 4 hh + j (the same for every row)
    if (tid < 2 * 4 * 4 * 2) {
        const int l = tid >> 5, wv = (tid >> 3) & 3, gate = (tid >> 1) & 3, h2 = tid & 1;
        f32x4 bv;
#pragma unroll
        for (int j = 0; j < 4; ++j) bv[j] = p.bias[l][gate * UH + member * 32 + wv * 8 + 4 * h2 + j];
        bias_s[tid] = bv;
    }
    // head: A fragment of the wave's four MFMAs -- column m = lane & 31 is target o = m (zero for m >= O), k = 4 hh + j is the wave's unit 4 hh + j
    {
        f32x4 wv = {0.0f, 0.0f, 0.0f, 0.0f};
        if (n < O) {
#pragma unroll
            for (int j = 0; j < 4; ++j) wv[j] = p.w_out[(size_t)n * UH + member * 32 + wave * 8 + 4 * hh + j];
        }
        wo_s[wave * 64 + lane] = wv;
    }

    // exchange buffer and input: descriptors as scalar tuples for the DMA asm
    const unsigned long long hx_addr = reinterpret_cast<unsigned long long>(p.hx);
    u32x4 hx_desc;
    hx_desc[0] = __builtin_amdgcn_readfirstlane((unsigned)hx_addr);
    hx_desc[1] = __builtin_amdgcn_readfirstlane((unsigned)(hx_addr >> 32) & 0xFFFFu);
    hx_desc[2] = __builtin_amdgcn_readfirstlane((unsigned)p.hx_bytes);
    hx_desc[3] = 0x00020000u;
    const unsigned long long xf_addr = reinterpret_cast<unsigned long long>(p.xfrag);
    u32x4 xf_desc;
    xf_desc[0] = __builtin_amdgcn_readfirstlane((unsigned)xf_addr);
    xf_desc[1] = __builtin_amdgcn_readfirstlane((unsigned)(xf_addr >> 32) & 0xFFFFu);
    xf_desc[2] = __builtin_amdgcn_readfirstlane((unsigned)p.xfrag_bytes);
    xf_desc[3] = 0x00020000u;
    const unsigned long long mb_addr = reinterpret_cast<unsigned long long>(p.maskbits);
    u32x4 mb_desc;                                               // the keep bits [n_tiles][T][unit 128], one word per (tile, step, unit)
    mb_desc[0] = __builtin_amdgcn_readfirstlane((unsigned)mb_addr);
    mb_desc[1] = __builtin_amdgcn_readfirstlane((unsigned)(mb_addr >> 32) & 0xFFFFu);
    mb_desc[2] = __builtin_amdgcn_readfirstlane((unsigned)((size_t)p.n_tiles * T * UH * sizeof(unsigned)));
    mb_desc[3] = 0x00020000u;
    const unsigned mwl_lds = (unsigned)reinterpret_cast<unsigned long long>(mwl) + (unsigned)(wave * 256);
    const unsigned mw_voff = (unsigned)((lane & 7) * 4);
    const unsigned mw_unit0 = (unsigned)((member * 32 + wave * 8) * sizeof(unsigned));
    // flags [cluster][set 2][layer 2][member wave 16]: epoch = slices published; exchange [cluster][set 2][kind 3][parity 2][16 KB],
    // kind 0 = h_1, 1 = h_1 masked, 2 = h_2
    unsigned* const flags_c = p.xflags + (size_t)cluster * 2 * 2 * NFL;
    auto ex_base = [&](int s, int kind, int par) -> unsigned { return (unsigned)(((((size_t)cluster * 2 + s) * 3 + kind) * 2 + par) * SET_BYTES); };
    const unsigned xb_lds = (unsigned)reinterpret_cast<unsigned long long>(xb);       // LDS byte addresses
    const unsigned h1b_lds = (unsigned)reinterpret_cast<unsigned long long>(h1b);
    const unsigned m1b_lds = (unsigned)reinterpret_cast<unsigned long long>(m1b);
    const unsigned h2b_lds = (unsigned)reinterpret_cast<unsigned long long>(h2b);

    // ---- do all members of this cluster really share an XCD?
    if (wave == 0) {
        unsigned spins = 0, v = 0u;
        while (true) {
            v = 0x10u | my_xcc;
            if (lane < GH) v = __hip_atomic_load(xcc_words + cluster * GH + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__all((int)(v != 0u))) break;
            if (++spins > SPIN_LIMIT || __hip_atomic_load(p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                if (lane == 0) {
                    ctl[0] = 1;
                    __hip_atomic_store(p.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        const int same = __all((int)((v & 0xFu) == my_xcc));
        if (lane == 0) ctl[3] = same;
    }
    __syncthreads();
    if (ctl[0] != 0) return;
    // The slices are handed over by WRITE-THROUGH (sc1) stores although all members of a cluster share an XCD's L2: the form the MI355X guide
    // lists as valid wherever the workgroups run, and with two tiles in flight per cluster the later acknowledgement costs nothing (1.181 vs
    // 1.183 ms per launch).  (Round 4 made this change in the belief that plain stores had caused this kernel's cold-start fault -- whole
    // 32-row tiles off on a fresh process's first launch; round 5 found the real cause, the mask words' asm load whose destination hipcc
    // copied in front of its wait -- see the note above look_issue_plain's use below and DESIGN.md 4.17 -- and that the hand-over had been
    // sound in both flavours.)
    // per-lane addresses of the hooks' loads, computed once: the look at a set's flags (+ the set's 2 * NFL words; all four quarter-waves
    // read the same sixteen), the mask words of this lane's four units (+ (tile * T + t) * 128 words)
    const ape_desc_t fl_desc = ape_make_desc(p.xflags, (unsigned)(NC * 2 * 2 * NFL * sizeof(unsigned)));
    const unsigned fl_off = (unsigned)(cluster * 2 * 2 * NFL * sizeof(unsigned));
    const unsigned look_voff = (unsigned)((lane & 15) * sizeof(unsigned));
    const unsigned look_lds = (unsigned)reinterpret_cast<unsigned long long>(look_s) + (unsigned)(wave * 256);
    const unsigned* const look_mine = look_s + wave * 64 + lane;
    // every wave polls for itself: have all member waves published epoch `want` of set s?
    auto wait_flags = [&](const unsigned* fl, unsigned want) {
        unsigned spins = 0;
        while (true) {
            unsigned v = want;
            if (lane < NFL) v = __hip_atomic_load(fl + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__all((int)(v >= want))) return;
            if (++spins > SPIN_LIMIT || __hip_atomic_load(p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                if (lane == 0) {
                    ctl[0] = 1;
                    __hip_atomic_store(p.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                return;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    };
    // a 16 KB copy global -> LDS: wave w moves KiB w, w + 4, w + 8, w + 12; piece k on its own so that a copy can be spread over k-blocks
    const unsigned dma_voff = (unsigned)(lane * 16);
    auto opaque = [](unsigned v) -> unsigned { asm volatile("" : "+s"(v)); return v; };     // (no hoisted address sums: lstm_upper32.hip)
    const unsigned wave_kib = (unsigned)(wave * 1024);
    // (`nrec` = the descriptor's record count: 0 turns the instruction into one that touches no memory -- zeros into an LDS buffer nobody
    //  reads before its real copy -- so that the hooks of a section need no branch on a verdict)
    auto copy_piece = [&](unsigned lds_base, int s, u32x4 desc, unsigned nrec, unsigned src, int k) {
        desc[2] = nrec;
        dma_1k(opaque(lds_base + wave_kib) + (unsigned)s * SET_BYTES + (unsigned)(k * 4096), dma_voff, desc, src + (unsigned)(wave * 1024 + k * 4096));
    };
    // the flag a wave owes for the slices it stored last (three stores, one flag): raised once those stores have drained
    // (address and value sit in vector registers from the publish on: no VALU instruction for them inside the next section's MFMA stream)
    bool pend = false;
    unsigned* pend_ptr = flags_c;
    unsigned pend_epoch = 0u;
    auto raise_pending = [&]() {              // caller has waited vmcnt(0)
        if (!pend) return;
        if (lane == 0) asm volatile("global_store_dword %0, %1, off sc1" :: "v"(pend_ptr), "v"(pend_epoch) : "memory");
        pend = false;
    };
    auto owe = [&](int idx, unsigned epoch) {
        pend = true;
        pend_ptr = flags_c + idx;
        pend_epoch = epoch;
        asm volatile("" : "+v"(pend_ptr), "+v"(pend_epoch));
    };
    auto bar = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    // ---- per-set state (uniform over the workgroup AND over the cluster: every member walks the same tiles).  A set's tiles are one
    //      stream of steps: section k of the set = layer 1 on step k of that stream AND layer 2 on step k - 1 (the last step of the tile
    //      in front when k is a tile's first), so a set with n tiles has n * T + 1 sections, the first without a layer-2 part, the last
    //      without a layer-1 part.
    float cst[2][2][4];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int l = 0; l < 2; ++l)
#pragma unroll
            for (int j = 0; j < 4; ++j) cst[s][l][j] = 0.0f;
    int a_tile[2], a_t[2], b_tile[2], b_t[2];    // the (tile, step) of the set's NEXT section: layer-1 part (a), layer-2 part (b)
    bool a_act[2], b_act[2];                     // ... and whether it has that part
    unsigned ksec[2] = {0u, 0u};                 // sections the set has run = slices epochs it has published
    bool pre[2] = {false, false};                // the operands of the set's next section are on their way (issued by the other set's section)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int tl = cluster + s * NC;
        a_act[s] = tl < p.n_tiles;
        a_tile[s] = tl; a_t[s] = 0;
        b_act[s] = false;
        b_tile[s] = tl; b_t[s] = 0;
    }
    const int frag = n * 8 + hh * 4;                              // this lane's 16 bytes inside a [row][8 units] block
    const unsigned pub_off = (unsigned)((((member * 4 + wave) * MR + n) * 8 + 4 * hh) * sizeof(float));
    const float keep = 1.0f / (1.0f - p.dropout_p);
    const unsigned xf_rec = xf_desc[2], hx_rec = hx_desc[2];
    constexpr unsigned NOWHERE = 0x80000000u;                     // a store offset beyond the exchange buffer: dropped by the bounds check

    // the sixteen 1-KiB pieces (per wave) of the operands of set s's NEXT section, piece i = 0 .. 15:
    //   0..3   layer 1's input: the pre-laid tile-step of p.xfrag                              (if the section has a layer-1 part)
    //   4..7   h_1 of the step before            (kind 0)   ... and that step is not a tile's first
    //   8..11  layer 2's input: h_1 of ITS step under the rows' masks (kind 1)                 (if it has a layer-2 part)
    //   12..15 h_2 of the step before            (kind 2)   ... and that step is not a tile's first
    // the three kinds were published together by the set's section in front (epoch ksec[s], parity of ksec[s] - 1)
    auto issue_piece = [&](int s, int i, bool go) {
        const int par = (int)((ksec[s] - 1u) & 1u);
        const int k = i & 3;
        if (i < 4) copy_piece(xb_lds, s, xf_desc, (go && a_act[s]) ? xf_rec : 0u, (unsigned)(((size_t)a_tile[s] * T + a_t[s]) * SET_BYTES), k);
        else if (i < 8) copy_piece(h1b_lds, s, hx_desc, (go && a_act[s] && a_t[s] != 0) ? hx_rec : 0u, ex_base(s, 0, par), k);
        else if (i < 12) copy_piece(m1b_lds, s, hx_desc, (go && b_act[s]) ? hx_rec : 0u, ex_base(s, 1, par), k);
        else copy_piece(h2b_lds, s, hx_desc, (go && b_act[s] && b_t[s] != 0) ? hx_rec : 0u, ex_base(s, 2, par), k);
    };

#ifdef APE_UP128_ASSERT
    // Asserting diagnostic build (tests/tools/assert_up128.py; never shipped): the tool loads weights under which a layer's fresh h is the
    // same number for every unit and row and depends on the step alone (W = 0, per-gate constant biases), so every float a section finds in
    // its gathered operands must equal what THIS lane computed in the set's section in front -- existing state, no tags, no extra traffic.
    // Wave 0 checks all 3 x 16 KB behind the top barrier (phase 0) and again at the section's end (phase 1: a copy that landed late shows
    // as bad in phase 0 only); a mismatch leaves a record {who, where, found, expected, clock} in dbg_wg[1024 ...].
    float chkA[2] = {0.0f, 0.0f}, chkB[2] = {0.0f, 0.0f};
    auto chk_record = [&](unsigned s_, unsigned k_, unsigned kind, unsigned kb, unsigned phase, float found, float expect) {
        const unsigned slot = __hip_atomic_fetch_add(reinterpret_cast<unsigned*>(p.dbg_wg + 1024), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (slot < 120u) {
            unsigned long long* o = p.dbg_wg + 1032 + slot * 4;
            o[0] = ((unsigned long long)cluster << 48) | ((unsigned long long)member << 40) | ((unsigned long long)s_ << 32) | k_;
            o[1] = ((unsigned long long)phase << 48) | ((unsigned long long)kind << 32) | (kb << 8) | (unsigned)lane;
            o[2] = ((unsigned long long)__builtin_bit_cast(unsigned, found) << 32) | __builtin_bit_cast(unsigned, expect);
            o[3] = __builtin_amdgcn_s_memtime();
        }
    };
#endif
#ifdef APE_CLUSTER_STAMPS
    // diagnostic counters and shader-clock sums (cluster 0, member 0, wave 0): sections, blocking tops, cycles in the top of a section
    // (wait + barrier), its MFMA chains, the gate math (+ head), the publish
    unsigned long long dg_block = 0, dg_go = 0, dg_sections = 0, dg_top = 0, dg_chain = 0, dg_gates = 0, dg_pub = 0, dg_t0 = 0;
#define UP_STAMP(acc) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); acc += now_ - dg_t0; dg_t0 = now_; }
#else
#define UP_STAMP(acc)
#endif
    // One section = one step of one set, both layers.  Vector-memory queue of a wave in a steady-state section, in issue order:
    //   [the three publish stores of the section in front]   QF: their flag; mask words   QP: flag look   QJ ..: the sixteen copies for the
    //   NEXT section (the other set's)   [head partial store]   three publish stores
    // so at the top everything but the three youngest entries is waited for, and a few k-blocks in those have drained too.
    auto section = [&](auto set_tag) -> bool {
        constexpr int s = decltype(set_tag)::value, o = s ^ 1;
        const unsigned k = ksec[s];
        const bool actA = a_act[s], actB = b_act[s];
        const int tileA = a_tile[s], tA = a_t[s], tileB = b_tile[s], tB = b_t[s];
        const bool lastA = tA == T - 1, lastB = tB == T - 1;
        const bool recA = actA && tA != 0, recB = actB && tB != 0;      // (h_{-1} = 0: a tile's first step has no recurrent span)
#ifdef APE_CLUSTER_STAMPS
        dg_sections += 1;
        dg_t0 = __builtin_amdgcn_s_memtime();
        if (!pre[s]) dg_block += 1;
#endif
        // ---- top: this section's operands
        if (pre[s]) {
            asm volatile("s_waitcnt vmcnt(3)" ::: "memory");      // the prefetched copies (+ a head partial store); only the publish stores are younger
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            raise_pending();
            if (k != 0u) wait_flags(flags_c + s * 2 * NFL, k);
#pragma unroll
            for (int i = 0; i < 4 * NDMA; ++i) issue_piece(s, i, true);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        pre[s] = false;
        bar();                                                      // (unconditional: lstm_upper32.hip on why)
        UP_STAMP(dg_top);
        const int abort_word = ctl[0];
#ifdef APE_UP128_ASSERT
        auto chk_all = [&](unsigned phase) {
            if (wave != 0 || p.dbg_wg == nullptr) return;
            const float* chk_base = smem;
            asm volatile("" : "+v"(chk_base));
            int bad = -1;                                          // first (kind, k-block) of this lane whose 16 bytes are not what they must be
            float bad_v = 0.0f;
#pragma unroll 1
            for (int idx = 0; idx < 3 * BH; ++idx) {
                const int kind = idx / BH, kb = idx - kind * BH;
                const bool on = (kind == 0) ? recA : (kind == 1) ? actB : recB;
                if (!on) continue;
                const float expect = (kind == 2) ? chkB[s] : chkA[s];
                // (h1b / m1b / h2b = smem + 2 HL, + 4 HL, + 6 HL; through a vector-register copy of the base, or the scalar LDS addresses
                //  of the copies end up in vector registers with it)
                const f32x4 v = *reinterpret_cast<const f32x4*>(chk_base + (2 + 2 * kind + s) * HL + frag + kb * (MR * 8));
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    // (the masked slice: 0 or keep x h_1, to an ulp -- the steps of a layer's sequence lie 1e-3 and more apart)
                    const bool fine = (kind == 1) ? (v[j] == 0.0f || __builtin_fabsf(v[j] - expect * keep) <= 1e-6f * __builtin_fabsf(v[j]))
                                                  : (__builtin_bit_cast(unsigned, v[j]) == __builtin_bit_cast(unsigned, expect));
                    if (!fine && bad < 0) { bad = idx; bad_v = v[j]; }
                }
            }
            if (bad >= 0) chk_record((unsigned)s, k, (unsigned)(bad / BH), (unsigned)(bad % BH), phase, bad_v, (bad / BH == 2) ? chkB[s] : chkA[s]);
        };
        chk_all(0u);
#endif
        // the NEXT section is the other set's (if it has one left)
        const bool o_more = a_act[o] || b_act[o];
        const unsigned want = ksec[o];
        unsigned peek = 0u;
        bool go = false;
        u32x4 mw = {0u, 0u, 0u, 0u};
        // hooks in the MFMA stream (k-block q of the section's 64, a constant after unrolling; a span that does not run -- no layer-1 /
        // layer-2 part, a tile's first step -- leaves its hooks behind, back to back):
        //   QF   the flag owed for the publish stores of the section in front (drained by now); the mask words of layer 1's outputs
        //   QP   look at the other set's flags      QJ   judge      QJ .. QJ + 15   one piece of the next section's operands per block
        constexpr int QF = UP128_QF, QP = UP128_QP, QJ = QP + UP128_QD;
        auto mid = [&](int q) {
            if (q == QF) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                raise_pending();
                // (no layer-1 part: tile 0's words, which nobody uses)
                look_issue_plain(mwl_lds, mw_voff, mb_desc, (unsigned)(((actA ? tileA : 0) * T + tA) * (UH * (int)sizeof(unsigned))) + mw_unit0);
            }
            if (q == QP) look_issue(look_lds, look_voff, fl_desc, fl_off + (unsigned)(o * 2 * NFL * sizeof(unsigned)));
            if (q == QJ - 1) {
                look_landed();                                    // (the mask words, requested at QF, have landed in LDS by now as well)
                peek = *look_mine;
                mw = *reinterpret_cast<const u32x4*>(mwl + wave * 64 + 4 * hh);
            }
            if (q == QJ) {
                go = o_more && (want == 0u || __builtin_amdgcn_ballot_w64(peek >= want) == ~0ull);
            }
            if (q >= QJ && q < QJ + 4 * NDMA) issue_piece(o, q - QJ, go);
        };
        auto hooks = [&](int q0) {
#pragma unroll
            for (int q = q0; q < q0 + BH; ++q) mid(q);
        };
        // ---- stacked-gate products: one dependent chain of 32x32x2 MFMAs per layer
        f32x16 accA, accB;
#pragma unroll
        for (int gate = 0; gate < 4; ++gate) {
            const f32x4 b1 = bias_s[((0 * 4 + wave) * 4 + gate) * 2 + hh], b2 = bias_s[((1 * 4 + wave) * 4 + gate) * 2 + hh];
            accA[4 * gate] = b1[0]; accA[4 * gate + 1] = b1[1]; accA[4 * gate + 2] = b1[2]; accA[4 * gate + 3] = b1[3];
            accB[4 * gate] = b2[0]; accB[4 * gate + 1] = b2[1]; accB[4 * gate + 2] = b2[2]; accB[4 * gate + 3] = b2[3];
        }
        if (actA) span32<BH, 2 * NWL>(accA, xb + s * HL + frag, MR * 8, w, 0, [&](int q) { mid(q); });
        else hooks(0);
        if (recA) span32<BH, 2 * NWL>(accA, h1b + s * HL + frag, MR * 8, w, 4 * BH, [&](int q) { mid(BH + q); });
        else hooks(BH);
        if (actB) span32<BH, 2 * NWL>(accB, m1b + s * HL + frag, MR * 8, w, NWL, [&](int q) { mid(2 * BH + q); });
        else hooks(2 * BH);
        if (recB) span32<BH, 2 * NWL>(accB, h2b + s * HL + frag, MR * 8, w, NWL + 4 * BH, [&](int q) { mid(3 * BH + q); });
        else hooks(3 * BH);
        // (the other set idle: THIS set runs the next section too, and its top copies into buffers read above -- every wave must be
        //  through with them first; `o_more` is state, uniform over the workgroup, so the extra barrier pairs up)
        if (!o_more) bar();
        mfma_drain(accA);
        mfma_drain(accB);
        UP_STAMP(dg_chain);
        // ---- gates + cell update, lane-local: registers 4 gate + j = gate of unit 4 hh + j, row n
        float hA[4] = {0.0f, 0.0f, 0.0f, 0.0f}, hB[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        auto gates = [&](const f32x16& acc, float (&c_)[4], float (&hn)[4]) {
#pragma unroll
            for (int j = 0; j < 4; j += 2) {
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                auto exp2_2 = [](f32x2 v) { return f32x2{__builtin_amdgcn_exp2f(v[0]), __builtin_amdgcn_exp2f(v[1])}; };
                auto rcp_2 = [](f32x2 v) { return f32x2{__builtin_amdgcn_rcpf(v[0]), __builtin_amdgcn_rcpf(v[1])}; };
                const f32x2 ai = {acc[j], acc[j + 1]}, af = {acc[4 + j], acc[5 + j]}, ag = {acc[8 + j], acc[9 + j]}, ao = {acc[12 + j], acc[13 + j]};
                const f32x2 iv = rcp_2(1.0f + exp2_2(-1.4426950408889634f * ai));
                const f32x2 fv = rcp_2(1.0f + exp2_2(-1.4426950408889634f * af));
                const f32x2 gv = 2.0f * rcp_2(1.0f + exp2_2(-2.885390081777927f * ag)) - 1.0f;
                const f32x2 ov = rcp_2(1.0f + exp2_2(-1.4426950408889634f * ao));
                const f32x2 c = fv * f32x2{c_[j], c_[j + 1]} + iv * gv;     // (a tile's first step finds c = 0: f * 0 + i * g = i * g exactly)
                c_[j] = c[0]; c_[j + 1] = c[1];
                const f32x2 h = ov * (2.0f * rcp_2(1.0f + exp2_2(-2.885390081777927f * c)) - 1.0f);
                hn[j] = h[0]; hn[j + 1] = h[1];
            }
        };
        if (actA) gates(accA, cst[s][0], hA);
        if (actB) gates(accB, cst[s][1], hB);
#ifdef APE_UP128_ASSERT
        chk_all(1u);
        chkA[s] = hA[0]; chkB[s] = hB[0];
        if (p.dbg_wg != nullptr && s == 0 && cluster == 0 && member == 0 && tid == 0 && k < 32u) {      // the value table: h_1, h_2 by section
            p.dbg_wg[1600 + 2 * k] = __builtin_bit_cast(unsigned, hA[0]);
            p.dbg_wg[1601 + 2 * k] = __builtin_bit_cast(unsigned, hB[0]);
        }
#endif
        if (abort_word != 0) return false;                        // (a wave of this workgroup gave up in a blocking wait)
        pre[o] = go;
#ifdef APE_CLUSTER_STAMPS
        if (go) dg_go += 1;
#endif
        if (actB && lastB) {
            // ---- head: partial y over this wave's 8 units = four more MFMAs, the fresh h_2 values are the activation fragment
            f32x16 ya;
#pragma unroll
            for (int i = 0; i < 16; ++i) ya[i] = 0.0f;
            const f32x4 wv = wo_s[wave * 64 + lane];
#pragma unroll
            for (int j = 0; j < 4; ++j) mfma32<false>(ya, wv[j], hB[j]);
            mfma_drain(ya);
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int j = 0; j < 4; ++j) hp[(wave * PO + 8 * g + 4 * hh + j) * MR + n] = ya[4 * g + j];
            bar();
            if (tid < 128) {
                const int rn = tid >> 2, oq = tid & 3;
                f32x4 sum;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int idx = (4 * oq + i) * MR + rn;
                    sum[i] = ((hp[idx] + hp[PO * MR + idx]) + hp[2 * PO * MR + idx]) + hp[3 * PO * MR + idx];
                }
                *reinterpret_cast<f32x4*>(p.ypart + (((size_t)tileB * MR + rn) * GH + member) * PO + 4 * oq) = sum;
            }
        }
        UP_STAMP(dg_gates);
        // ---- publish: a lane's four fresh values of a layer are one 16-byte piece of the exchange layout.  Always THREE stores (the
        //      counted wait at the next top relies on it): h_1 plain (layer 1's recurrence), h_1 under the rows' masks (layer 2's input),
        //      h_2 plain; what nobody will read -- a tile's last step's plain values, a part the section did not have -- goes to an
        //      offset beyond the buffer and is dropped.
        {
            const int par = (int)(k & 1u);
            const u32x4 v1 = {__builtin_bit_cast(unsigned, hA[0]), __builtin_bit_cast(unsigned, hA[1]),
                              __builtin_bit_cast(unsigned, hA[2]), __builtin_bit_cast(unsigned, hA[3])};
            const u32x4 v2 = {__builtin_bit_cast(unsigned, hB[0]), __builtin_bit_cast(unsigned, hB[1]),
                              __builtin_bit_cast(unsigned, hB[2]), __builtin_bit_cast(unsigned, hB[3])};
            u32x4 vm;
#pragma unroll
            for (int j = 0; j < 4; ++j) vm[j] = __builtin_bit_cast(unsigned, ((mw[j] >> n) & 1u) ? hA[j] * keep : 0.0f);
            const unsigned off0 = (actA && !lastA) ? ex_base(s, 0, par) + pub_off : NOWHERE;
            const unsigned off1 = actA ? ex_base(s, 1, par) + pub_off : NOWHERE;
            const unsigned off2 = (actB && !lastB) ? ex_base(s, 2, par) + pub_off : NOWHERE;
            store_16(v1, off0, hx_desc); store_16(vm, off1, hx_desc); store_16(v2, off2, hx_desc);
            owe(s * 2 * NFL + member * 4 + wave, k + 1u);
        }
        // ---- the set's next section: layer 2 follows layer 1 one step behind; layer 1 moves on, to the set's next tile behind a last step
        ksec[s] = k + 1u;
        b_act[s] = actA; b_tile[s] = tileA; b_t[s] = tA;
        if (actA && tA == 0) {                                    // layer 2 starts that tile next: c_2 = 0
#pragma unroll
            for (int j = 0; j < 4; ++j) cst[s][1][j] = 0.0f;
        }
        if (actA) {
            if (lastA) {
                const int nt = tileA + 2 * NC;
                a_act[s] = nt < p.n_tiles;
                a_tile[s] = nt; a_t[s] = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) cst[s][0][j] = 0.0f;
            } else {
                a_t[s] = tA + 1;
            }
        }
        UP_STAMP(dg_pub);
        return true;
    };

    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
#ifdef APE_CLUSTER_STAMPS
    const unsigned long long dg_first = __builtin_amdgcn_s_memtime();
#endif
    bool ok = true;
#pragma unroll 1
    while (ok && (a_act[0] || b_act[0] || a_act[1] || b_act[1])) {
        if (a_act[0] || b_act[0]) ok = section(S0{});
        if (ok && (a_act[1] || b_act[1])) ok = section(S1{});
    }
    if (!ok) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // (a flag still owed is awaited by nobody: dropped, lstm_upper32.hip)

#ifdef APE_CLUSTER_STAMPS
    if (p.dbg_wg != nullptr && tid == 0 && cluster == 0 && member == 0) {
        p.dbg_wg[16] = dg_block; p.dbg_wg[17] = dg_go; p.dbg_wg[18] = dg_sections;
        p.dbg_wg[19] = dg_top; p.dbg_wg[20] = dg_chain; p.dbg_wg[21] = dg_gates; p.dbg_wg[22] = dg_pub;
    }
    if (p.dbg_wg != nullptr && tid == 0 && member == 0 && cluster < 64) {      // per cluster: entry, first section, exit (shader clock)
        p.dbg_wg[64 + cluster] = dg_kstart; p.dbg_wg[128 + cluster] = dg_first; p.dbg_wg[192 + cluster] = __builtin_amdgcn_s_memtime();
    }
#endif
    // ---- self-cleaning: the last workgroup out re-zeroes every polled word
    __syncthreads();
    if (tid == 0)
        ctl[2] = (__hip_atomic_fetch_add(p.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) ? 1 : 0;
    __syncthreads();
    if (ctl[2] != 0) {
        const int n_flags = NC * 2 * 2 * NFL;
        for (int i = tid; i < n_flags; i += 256) __hip_atomic_store(p.xflags + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int i = tid; i < (int)gridDim.x; i += 256) __hip_atomic_store(xcc_words + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid < 8) __hip_atomic_store(class_ticket + tid * 16, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid == 0) __hip_atomic_store(p.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

constexpr size_t smem_upper128() {
    return (size_t)8 * HL * sizeof(float) + (size_t)(2 * 4 * 4 * 2 + 4 * 64) * 16 + (size_t)4 * PO * MR * sizeof(float) + (size_t)2 * 4 * 64 * sizeof(unsigned) + 16;
}

// ---- launch B's inputs: layer 0's output under each sample row's mask in fragment order [tile][step][k-block 16][row 32][8 units], and
//      the keep bits of layer 1's outputs [tile][step][unit 128] (bit n = row n of the tile).  Sample row r (row_base + its index in this
//      chunk) is sample r % n_mc of stream r / n_mc.  Philox with the counters every fused kernel uses (rows r & ~3, step, unit, layer;
//      value index r & 3 -- lstm_tile16.hip), so the samples are the ones the batch-tile route draws.  One workgroup (128 threads = units)
//      per (tile, step).
constexpr int XS = 8 * MR + 8;           // LDS stride of a k-block (floats)

__global__ __launch_bounds__(128) void ape_mc_expand128_kernel(const ExpandParams q, unsigned* __restrict__ maskbits) {
    __shared__ __attribute__((aligned(16))) float sl[BH * XS];
    const int unit = threadIdx.x;
    const unsigned tile = blockIdx.x / (unsigned)q.T, t = blockIdx.x - tile * (unsigned)q.T;
    const unsigned row0 = tile * MR;                              // first row of the tile, chunk-local
    const float keep = 1.0f / (1.0f - q.dropout_p);
    const bool drop = q.dropout_p > 0.0f;
    const unsigned g0 = (unsigned)q.row_base + row0;              // global index of the tile's first row (< 2^31)
    unsigned stream = g0 / (unsigned)q.n_mc, rem = g0 - stream * (unsigned)q.n_mc;
    unsigned bits1 = 0u;
#pragma unroll
    for (int g = 0; g < MR / 4; ++g) {
        const unsigned r4 = g0 + 4 * g;                           // global index of the row quad (a multiple of 4)
        uint32_t rnd[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, rn1[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
        if (drop && q.masks == nullptr) {
            philox4x32((uint32_t)r4, (uint32_t)t, (uint32_t)unit, (uint32_t)q.layer, (uint32_t)q.seed, (uint32_t)(q.seed >> 32), rnd);
            philox4x32((uint32_t)r4, (uint32_t)t, (uint32_t)unit, (uint32_t)(q.layer + 1), (uint32_t)q.seed, (uint32_t)(q.seed >> 32), rn1);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float v = 0.0f;
            bool keep1 = true;
            const bool live = row0 + 4 * g + i < (unsigned)q.rows;
            if (live) {
                v = q.hseq[((size_t)stream * q.T + t) * UH + unit];
                if (q.masks != nullptr) {          // the caller's multipliers (test hooks: the bank against the oracle under the same masks);
                    // layer 1's output mask travels as a keep bit: the multiplier is 0 or 1 / (1 - p), the kernel's own `keep`
                    v *= q.masks[(((size_t)q.layer * q.masks_rows + (r4 + i)) * q.T + t) * UH + unit];
                    keep1 = q.masks[(((size_t)(q.layer + 1) * q.masks_rows + (r4 + i)) * q.T + t) * UH + unit] != 0.0f;
                } else if (drop) {
                    const float uf = (float)(rnd[i] >> 8) * (1.0f / 16777216.0f);
                    v = (uf >= q.dropout_p) ? v * keep : 0.0f;
                }
            }
            const float uf1 = (float)(rn1[i] >> 8) * (1.0f / 16777216.0f);
            if (q.masks != nullptr ? keep1 : (!drop || uf1 >= q.dropout_p)) bits1 |= 1u << (4 * g + i);
            sl[(unit >> 3) * XS + (4 * g + i) * 8 + (unit & 7)] = v;
            if (++rem == (unsigned)q.n_mc) { rem = 0u; ++stream; }
        }
    }
    maskbits[((size_t)tile * q.T + t) * UH + unit] = bits1;
    __syncthreads();
    f32x4* dst = reinterpret_cast<f32x4*>(q.xfrag + ((size_t)tile * q.T + t) * HL);
#pragma unroll
    for (int e = 0; e < HL / 4 / 128; ++e) {
        const int idx = threadIdx.x + 128 * e;                    // float4 index: k-block idx / 64, inside it idx % 64
        dst[idx] = *reinterpret_cast<const f32x4*>(sl + (idx >> 6) * XS + (idx & 63) * 4);
    }
}

// y[r][o] = b_out[o] + the four members' partial sums, member 0 first (a fixed order: run-to-run identical bits)
__global__ __launch_bounds__(256) void ape_head_reduce128_kernel(const float* __restrict__ ypart, const float* __restrict__ b_out,
                                                                 float* __restrict__ y, int rows, int O) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * PO) return;
    const int r = idx / PO, o = idx - r * PO;
    if (o >= O) return;
    const float* src = ypart + (size_t)r * GH * PO + o;
    float s = src[0];
#pragma unroll
    for (int m = 1; m < GH; ++m) s += src[m * PO];
    y[(size_t)r * O + o] = s + b_out[o];
}

}  // namespace

bool ape_upper128_supported(int H, int L, int O) { return H == UH && L == 3 && O <= PO; }
size_t ape_upper128_xfrag_bytes(int rows, int T) { return (size_t)((rows + MR - 1) / MR) * T * SET_BYTES; }
size_t ape_upper128_maskbits_bytes(int rows, int T) { return (size_t)((rows + MR - 1) / MR) * T * UH * sizeof(unsigned); }
size_t ape_upper128_ypart_bytes(int rows) { return (size_t)((rows + MR - 1) / MR) * MR * GH * PO * sizeof(float); }
size_t ape_upper128_hx_bytes(int clusters) { return (size_t)clusters * (2 * 3 * 2) * SET_BYTES; }
size_t ape_upper128_flag_words(int clusters) { return (size_t)clusters * 2 * 2 * NFL; }

hipError_t ape_prepare_lstm_upper128() {
    static_assert(smem_upper128() <= APE_LDS_BYTES, "LDS layout exceeds a CU");
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&ape_lstm_upper128), hipFuncAttributeMaxDynamicSharedMemorySize, APE_LDS_BYTES);
}

// one chunk of sample rows: expand -> layers 1 and 2 -> head reduce, all on `stream`.  `max_clusters` = 4-member clusters the device holds
// at once (a multiple of 8: whole block-index classes); the grid is the smaller of that and the tiles, rounded up to 8.
hipError_t ape_launch_lstm_upper128(const Upper128Params& p, const ExpandParams& q, const float* b_out, float* y, int max_clusters,
                                    hipStream_t stream, hipEvent_t ev_begin, hipEvent_t ev_end) {
    if (p.n_tiles < 1 || max_clusters < 8 || p.O > PO) return hipErrorInvalidValue;
    hipLaunchKernelGGL(ape_mc_expand128_kernel, dim3(p.n_tiles * p.T), dim3(128), 0, stream, q, const_cast<unsigned*>(p.maskbits));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    int clusters = (p.n_tiles + 7) / 8 * 8;
    if (clusters > max_clusters) clusters = max_clusters;
    if (ev_begin) (void)hipEventRecord(ev_begin, stream);          // (measurement aid: ape_streams_profile)
    hipLaunchKernelGGL(ape_lstm_upper128, dim3(clusters * GH), dim3(256), smem_upper128(), stream, p);
    e = hipGetLastError();
    if (ev_end) (void)hipEventRecord(ev_end, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(ape_head_reduce128_kernel, dim3((q.rows * PO + 255) / 256), dim3(256), 0, stream, p.ypart, b_out, y, q.rows, p.O);
    return hipGetLastError();
}
