// Weight-stationary UPPER-layer LSTM kernel for the Monte-Carlo stream bank (the estimators' default mode: n_mc dropout
// samples per stream and frame, reference estimate/nn_models.py:191-207 + watch_phone_pocket_nn.py:13-19), exact float32.
//
// nn.LSTM's dropout sits between the layers (nn_models.py:169-174), so layer 0 is computed once per stream (launch A of
// ape_streams_step) and the layer above runs over the S x n_mc sample rows as a ONE-layer LSTM whose 256-wide input is that
// sequence under each sample's own mask.  Until round 3 this launch ran on the batch-tile kernel (weights re-streamed from L2
// for every 16-row tile: 65 % of the f32 MFMA peak, 94 % of a bank frame).  Here:
//
//   * one layer with K = 256 + 256 is exactly the 256 accumulator-file registers per lane of an 8-member cluster of
//     lstm_cluster32.hip (a member = workgroup = CU owns 32 hidden units, a wave 8 of them x 4 gates = the 32 columns of one
//     v_mfma_f32_32x32x2_f32 tile; the weights are the A operand and never move after the prologue);
//   * the kernel is PERSISTENT over row tiles: a cluster owns 32-row tiles c, c + NC, c + 2 NC, ... -- one weight prologue per
//     launch, not per 1024 rows;
//   * two tiles are in flight per cluster ("sets" 0 / 1, their steps alternate): while set s computes, the slices set s^1
//     published at the end of its step travel (store -> acknowledged -> flag -> look -> LDS-DMA gather), and so does s^1's next
//     input tile -- the exchange of a recurrence step has a whole section (16.4K MFMA cycles) to hide in, and the sets share
//     nothing (no second dependent round trip, as in lstm_cluster_f16v2.hip);
//   * the masked input arrives pre-laid in MFMA fragment order [tile][step][k-block 32][window 32][8 units] (written by
//     ape_mc_expand_kernel below with the Philox counters of the fused kernels, so the samples are bit-for-bit the ones the
//     batch-tile route draws): 32 LDS-DMA instructions per workgroup and step copy it global -> LDS, no register, no VALU
//     instruction in the MFMA stream (a wave's own VALU work adds its issue time to a dependent MFMA chain on gfx950,
//     DESIGN.md 4.10);
//   * h_{-1} = 0: step 0 of a tile is the input span alone (the batch-tile kernel skips it too);
//   * the head (Linear(H, O) on the last step) needs no exchange: each wave multiplies its own 8 fresh units by its slice of
//     W_out with four more MFMAs (the fresh h values ARE the activation fragment), the four waves' partial sums meet in LDS,
//     and every member writes its 32-unit partial [row][member][16]; ape_head_reduce_kernel adds the eight in a fixed order.
// Exchange protocol, cluster formation (arrival tickets within the block-index class = XCD, verified at run time), bounded
// spins, sticky status word and self-cleaning are those of lstm_cluster32.hip.
#include <type_traits>

#include "ape_internal.h"
#include "async_look.h"
#include "../../include/ape_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((__vector_size__(4 * sizeof(unsigned))));
constexpr unsigned SPIN_LIMIT = 1u << 22;

__device__ __forceinline__ float sigm(float v) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v)); }
__device__ __forceinline__ float tanh_(float v) { return 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.885390081777927f * v)) - 1.0f; }

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

// A flag look that does NOT stall the MFMA stream: hipcc hoists the comparison of a compiler-visible load up to the load and puts
// `s_waitcnt vmcnt(0)` right behind it (an L2 round trip exposed in every section: found in the disassembly of round 2's kernels).
// The look is issued as LDS-DMA into the wave's landing zone (async_look.h: no destination register for hipcc to copy or re-use while the
// load is in flight -- round 5), read back from LDS one k-block in front of the judge.

// one LDS-DMA wave-instruction: 64 lanes x 16 bytes from the buffer (lane offset `voff`, uniform `soff`) to 1 KiB of LDS at
// the wave-uniform byte address `lds_addr`; sc1 = L1-bypassing, like every load of handed-off bytes
// the publish store through the SAME scalar descriptor tuple as the gathers (a second, compiler-built copy of it costs four more
// scalar registers in a kernel that already spills them); `sc1` = write-through form for clusters that span XCDs
template <bool WT>
__device__ __forceinline__ void store_16(u32x4 v, unsigned voff, u32x4 rsrc) {
    if constexpr (WT) asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen sc1" APE_STORE_TAIL :: "v"(v), "v"(voff), "s"(rsrc) : "memory");
    else asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen" APE_STORE_TAIL :: "v"(v), "v"(voff), "s"(rsrc) : "memory");
}
__device__ __forceinline__ void dma_1k(unsigned lds_addr, unsigned voff, u32x4 rsrc, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen sc1 lds"
                 :: "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

constexpr int UH = 256;                  // hidden units = input width of the layer
constexpr int GH = 8;                    // members per cluster
constexpr int MR = 32;                   // windows (sample rows) per tile
constexpr int BH = UH / 8;               // k-blocks of 8 per span
// (weight registers per lane: 4 * (KXB + BH) -- [W_ih | W_hh], all in the accumulator file; KXB = k-blocks of the input)
constexpr int NFL = 4 * GH;              // flags per (cluster, set): one per member wave
constexpr int HL = GH * 4 * MR * 8;      // floats of one slice set / one input tile-step [k-block 32][window 32][8] = 32 KB
constexpr int NDMA = 8;                  // LDS-DMA instructions per wave and 32 KB copy
constexpr int PO = 16;                   // width of a head partial row (O <= 16)
constexpr unsigned SET_BYTES = HL * sizeof(float);

// KXB = k-blocks of 8 of the layer's input: 32 (the layer above, 256-wide input) or 4 (layer 0 of the 2 x 256 models, I <= 32).
// SEQ: the layer-0 form -- no head; every step's slices go to p.hseq ([tile][step][32 KB], the fragment order the layer above's
// input builder reads) instead of the parity buffers, and the gather of step t reads step t - 1 from there.
template <int KXB, bool SEQ>
__global__ __launch_bounds__(256, 1) void ape_lstm_upper32(const UpperParams p) {
    constexpr int NW = 4 * (KXB + BH);
    constexpr int NXD = (KXB + 3) / 4;       // LDS-DMA instructions per wave and input tile (1 KiB each, 4 waves)
    constexpr unsigned XT_BYTES = KXB * 1024u;                   // bytes of one input tile-step
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, hh = lane >> 5;     // window of the tile, half (units 4 hh .. 4 hh + 3 of the wave's 8)
    const int T = p.T, O = SEQ ? 0 : p.O;
    // timing-only switches of the ablation library (make ablate; results are garbage): what does each part of a section cost?
#ifdef APE_ABLATE
    const unsigned ab = p.flags;
#else
    constexpr unsigned ab = 0u;
#endif
    const bool ab_noex = (ab & APE_DIAG_NO_EXCHANGE) != 0, ab_noact = (ab & APE_DIAG_NO_ACT) != 0, ab_nox = (ab & APE_DIAG_NO_XSTAGE) != 0,
               ab_nomfma = (ab & APE_DIAG_NO_MFMA) != 0, ab_nobar = (ab & APE_DIAG_NO_BARRIER) != 0;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xb = smem;                            // [set 2][HL]  masked input of the set's current step, fragment order
    float* hb = xb + 2 * HL;                     // [set 2][HL]  h_{t-1} of the set's tile, all members' slices
    f32x4* bias_s = reinterpret_cast<f32x4*>(hb + 2 * HL);        // [wave 4][gate 4][hh 2]: accumulator start values (b_ih + b_hh)
    f32x4* wo_s = bias_s + 4 * 4 * 2;                             // [wave 4][lane 64]: W_out as the head MFMAs' A fragment
    float* hp = reinterpret_cast<float*>(wo_s + 4 * 64);          // [wave 4][PO][MR]: head partial sums of the four waves
    unsigned* look_s = reinterpret_cast<unsigned*>(hp + 4 * PO * MR);   // [wave 4][64]: landing zones of the flag looks (async_look.h)
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

    // ---- weights: 256 registers per lane in the accumulator file, for the whole launch.  Host layout (ape_api.hip, wcl32):
    //      [member][wave][register / 4][lane][4]; register 4 kb + j of lane (column m = lane & 31 = gate * 8 + unit, half hh)
    //      = [W_ih | W_hh][gate * H + member * 32 + wave * 8 + unit][8 kb + 4 hh + j]
    float w[NW];
    {
        const f32x4* s1 = reinterpret_cast<const f32x4*>(p.w) + ((size_t)(member * 4 + wave) * (NW / 4)) * 64 + lane;
#pragma unroll
        for (int i = 0; i < NW / 4; ++i) {
            const f32x4 v = s1[i * 64];
            w[4 * i] = v[0]; w[4 * i + 1] = v[1]; w[4 * i + 2] = v[2]; w[4 * i + 3] = v[3];
        }
    }
    // accumulator start values: registers 4 gate + j <-> unit member*32 + wave*8 + 4 hh + j (the same for every window)
    if (tid < 4 * 4 * 2) {
        const int wv = tid >> 3, gate = (tid >> 1) & 3, h2 = tid & 1;
        f32x4 bv;
#pragma unroll
        for (int j = 0; j < 4; ++j) bv[j] = p.bias[gate * UH + member * 32 + wv * 8 + 4 * h2 + j];
        bias_s[tid] = bv;
    }
    // head: A fragment of the wave's four MFMAs -- column m = lane & 31 is target o = m (zero for m >= O), k = 4 hh + j is the
    // wave's unit 4 hh + j
    if constexpr (!SEQ) {
        f32x4 wv = {0.0f, 0.0f, 0.0f, 0.0f};
        if (n < O) {
#pragma unroll
            for (int j = 0; j < 4; ++j) wv[j] = p.w_out[(size_t)n * UH + member * 32 + wave * 8 + 4 * hh + j];
        }
        wo_s[wave * 64 + lane] = wv;
    }

    // exchange buffer and input: descriptors as scalar tuples for the DMA asm, and one for the compiler's publish store
    // (ONE exchange buffer per form: the parity slices p.hx, or -- SEQ -- the sequence p.hseq that is both exchange and output)
    float* const ex_ptr = SEQ ? p.hseq : p.hx;
    const size_t ex_bytes = SEQ ? p.hseq_bytes : p.hx_bytes;
    const unsigned long long hx_addr = reinterpret_cast<unsigned long long>(ex_ptr);
    u32x4 hx_desc;
    hx_desc[0] = __builtin_amdgcn_readfirstlane((unsigned)hx_addr);
    hx_desc[1] = __builtin_amdgcn_readfirstlane((unsigned)(hx_addr >> 32) & 0xFFFFu);
    hx_desc[2] = __builtin_amdgcn_readfirstlane((unsigned)ex_bytes);
    hx_desc[3] = 0x00020000u;
    const unsigned long long xf_addr = reinterpret_cast<unsigned long long>(p.xfrag);
    u32x4 xf_desc;
    xf_desc[0] = __builtin_amdgcn_readfirstlane((unsigned)xf_addr);
    xf_desc[1] = __builtin_amdgcn_readfirstlane((unsigned)(xf_addr >> 32) & 0xFFFFu);
    xf_desc[2] = __builtin_amdgcn_readfirstlane((unsigned)p.xfrag_bytes);
    xf_desc[3] = 0x00020000u;
    unsigned* const flags_of = p.xflags + (size_t)cluster * 2 * NFL;         // [set][member*4 + wave] epoch = slices published
    const ape_desc_t fl_desc = ape_make_desc(p.xflags, (unsigned)((gridDim.x / GH) * 2 * NFL * sizeof(unsigned)));
    const unsigned fl_off = (unsigned)(cluster * 2 * NFL * sizeof(unsigned));
    const unsigned look_voff = (unsigned)((lane & (NFL - 1)) * sizeof(unsigned));
    const unsigned look_lds = (unsigned)reinterpret_cast<unsigned long long>(look_s) + (unsigned)(wave * 256);
    const unsigned* const look_mine = look_s + wave * 64 + lane;
    auto hx_base = [&](int s, int par) -> unsigned { return (unsigned)((((size_t)cluster * 2 + s) * 2 + par) * SET_BYTES); };
    const unsigned xb_lds = (unsigned)reinterpret_cast<unsigned long long>(xb);       // LDS byte addresses
    const unsigned hb_lds = (unsigned)reinterpret_cast<unsigned long long>(hb);

    // ---- do all members of this cluster really share an XCD? ------------------------------------------------------
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
    // The hand-over is by write-through (sc1) stores whatever the placement: the form the MI355X guide lists as valid wherever the workgroups
    // run, and with two tiles in flight per cluster the later acknowledgement is free (tests/tools/ab_wt.py: 1202.1 vs 1200.7 us per frame of
    // the 1024 x 25 bank, 1626.2 vs 1621.3 us of the watch-only bank).  (Round 4 made this change in the belief that plain stores had caused
    // lstm_upper128.hip's cold-start fault; round 5 found the real cause -- an asm load's destination copied in front of its wait,
    // async_look.h -- and that the hand-over had been sound in both flavours: DESIGN.md 4.17.)
    constexpr bool in_l2 = false;

    // every wave polls for itself: have all member waves published epoch `want` of set s?
    auto wait_flags = [&](int s, unsigned want) {
        unsigned spins = 0;
        while (true) {
            unsigned v = want;
            if (lane < NFL) v = __hip_atomic_load(flags_of + s * NFL + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
    // a 32 KB copy global -> LDS: wave w moves KiB w, w + 4, ... (8 of them); piece k on its own so that a copy can be spread
    // over the k-blocks of a span
    const unsigned dma_voff = (unsigned)(lane * 16);
    // the slices published as epoch `epoch` (>= 1) of set s -> hb[s]; SEQ: they are step t - 1 of the set's tile in the sequence buffer
    // (the LDS base is laundered through an empty asm at every use: hipcc otherwise hoists all sixteen `base + k * 4096` sums of a
    //  section out of the loop, runs out of scalar registers and parks them in VGPR lanes -- a v_readlane in front of every copy)
    auto opaque = [](unsigned v) -> unsigned { asm volatile("" : "+s"(v)); return v; };
    const unsigned wave_kib = (unsigned)(wave * 1024);
    auto issue_h_piece = [&](int buf, int s, unsigned epoch, int tile, int t, int k) {      // -> hb[buf]; `s` names the set's exchange slots
        const unsigned dst = opaque(hb_lds + wave_kib) + (unsigned)buf * SET_BYTES + (unsigned)(k * 4096);
        if constexpr (SEQ) {
            dma_1k(dst, dma_voff, hx_desc, (unsigned)(((size_t)tile * T + (t - 1)) * SET_BYTES) + (unsigned)(wave * 1024 + k * 4096));
        } else {
            dma_1k(dst, dma_voff, hx_desc, hx_base(s, (int)((epoch - 1u) & 1u)) + (unsigned)(wave * 1024 + k * 4096));
        }
    };
    auto issue_x_piece = [&](int s, int tile, int t, int k) {     // input of (tile, step t) -> xb[s]: KiB wave + 4 k of its KXB
        if (wave + 4 * k >= KXB) return;
        const unsigned src = (unsigned)(((size_t)tile * T + t) * XT_BYTES) + (unsigned)(wave * 1024 + k * 4096);
        dma_1k(opaque(xb_lds + wave_kib) + (unsigned)s * SET_BYTES + (unsigned)(k * 4096), dma_voff, xf_desc, src);
    };
    // the flag a wave owes for the slice it stored last: raised once that store has drained
    int pend_idx = -1;
    unsigned pend_epoch = 0u;
    auto raise_pending = [&]() {              // caller has waited vmcnt(0)
        if (pend_idx < 0) return;
        if (lane == 0) __hip_atomic_store(flags_of + pend_idx, pend_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pend_idx = -1;
    };
    // workgroup barrier that waits for this wave's LDS traffic only (not for the publish store or a DMA in flight)
    auto bar = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    // ---- per-set state (uniform over the workgroup AND over the cluster: every member walks the same tiles) ------------------
    float cst[2][4];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j) cst[s][j] = 0.0f;
    int tile_of[2], step_of[2];              // the set's current tile (-1: none left) and step
    unsigned pub[2] = {0u, 0u};              // slices the set has published so far = epoch of its newest ones
    bool prex[2] = {false, false};           // the set's input tile / gathered slices for its NEXT section are on their way
    bool preh[2] = {false, false};           //   (issued by the other set's section)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int tl = cluster + s * NC;
        tile_of[s] = tl < p.n_tiles ? tl : -1;
        step_of[s] = 0;
    }
    const int frag = n * 8 + hh * 4;                              // this lane's 16 bytes inside a [window][8 units] block
    const unsigned pub_off = (unsigned)((((member * 4 + wave) * MR + n) * 8 + 4 * hh) * sizeof(float));
#ifdef APE_CLUSTER_STAMPS
    // diagnostic counters and shader-clock sums (cluster 0, member 0, wave 0): sections, blocking tops by cause, cycles in the
    // top of a section (wait + barrier), its MFMA chain, its gates + publish (+ head)
    unsigned long long dg_block_x = 0, dg_block_h = 0, dg_sections = 0, dg_top = 0, dg_chain = 0, dg_tail = 0, dg_t0 = 0;
#define UP_STAMP(acc) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); acc += now_ - dg_t0; dg_t0 = now_; }
#else
#define UP_STAMP(acc)
#endif

    // One section = one step of one set: S0 the set's operands are in LDS (prefetched by the other set's section, or fetched
    // here in the blocking form: pipeline fill, a late peer, the other set idle), barrier, input span (+ recurrent span for
    // t >= 1) with the OTHER set's traffic hung into the MFMA stream, gates, publish (or, last step, the head).
    // Vector-memory queue of a wave in a steady-state section, in issue order:
    //   [publish store of the section in front]  x DMA for the other set (8)  flag look (1 load)  h DMA for the other set (8)
    //   [head partial store, last step only]  publish store (1; a store to nowhere on a last step)
    // so at the top of a section everything but the youngest entry is waited for (`vmcnt(1)`), and a few k-blocks in the
    // store itself has drained and its flag goes up.
    // SOLO (round 5; the wide-input forms only): a cluster that owns ONE tile has no other set to hide its exchange behind -- every step paid
    // store -> acknowledge -> flag -> look -> gather in the open (2.6 us of 9.4 per step: ImuPoseLSTM at 1024 windows, one tile per cluster
    // and layer).  With a 32-block input span in front of the recurrent one the section hides it itself, as lstm_cluster32.hip's layer 1
    // does (MODE 3): the flag owed goes up at block QF, the look at the OWN set's flags and the gather of h_{t-1} ride
    // under the input span (which does not read them), one barrier between the spans; the next step's input tile is copied under the
    // same span.  LDS buffers alternate by step parity (the idle set's are free): what a section copies into was last read one section
    // earlier, behind this section's top barrier -- no barrier at the end.
    auto section = [&](auto set_tag, auto first_tag, auto solo_tag) -> bool {
        constexpr int s = decltype(set_tag)::value, o = s ^ 1;
        constexpr bool first = decltype(first_tag)::value;          // step 0 of a tile: no recurrent span (h_{-1} = 0)
        constexpr bool solo = decltype(solo_tag)::value;
        static_assert(!solo || (KXB >= 28 && s == 0), "the solo form needs the long input span");
        const int t = step_of[s], tile = tile_of[s];
        const bool last = t == T - 1;
        const int bx = solo ? (t & 1) : s, bh = bx;                // LDS buffers of this section's input tile / gathered slices
#ifdef APE_CLUSTER_STAMPS
        dg_sections += 1;
        dg_t0 = __builtin_amdgcn_s_memtime();
#endif
        // ---- S0 ------------------------------------------------------------------------------------------------------------------
        if (ab_nobar) {
        } else if (solo) {
            if (prex[s]) {
                asm volatile("s_waitcnt vmcnt(1)" ::: "memory");  // the input tile copied under the section in front; only its publish store is younger
            } else {                                               // the tile's first section
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                raise_pending();
#pragma unroll
                for (int k = 0; k < NXD; ++k) issue_x_piece(bx, tile, t, k);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        } else if ((prex[s] || ab_nox) && (first || preh[s] || ab_noex)) {
            asm volatile("s_waitcnt vmcnt(1)" ::: "memory");      // the prefetched copies; only the publish store is younger
        } else {
#ifdef APE_CLUSTER_STAMPS
            if (!prex[s]) dg_block_x += 1; else dg_block_h += 1;
#endif
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            raise_pending();
            if (!prex[s] && !ab_nox) {
#pragma unroll
                for (int k = 0; k < NXD; ++k) issue_x_piece(s, tile, t, k);
            }
            if (!first && !ab_noex) {
                wait_flags(s, pub[s]);
#pragma unroll
                for (int k = 0; k < NDMA; ++k) issue_h_piece(s, s, pub[s], tile, t, k);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        prex[s] = false;
        preh[s] = false;
        // (UNCONDITIONAL: each wave judged the peers' flags for itself, so the four waves may take different forms of S0 -- a
        //  barrier inside the blocking form would pair with the wrong one.  Round 3 tried the barrier at the end of the MFMA
        //  chain with none here for prefetched operands: no gain, and exactly that mis-pairing under rocprofv3's timing.)
        if (!ab_nobar) bar();
        const int abort_word = ctl[0];
        // the other set's next section: (tile_of[o], step_of[o]); it needs its input tile, and from step 1 on the slices it
        // published last (epoch pub[o])
        const bool o_act = !solo && tile_of[o] >= 0 && !ab_nox;
        const bool o_h = !solo && tile_of[o] >= 0 && step_of[o] >= 1 && !ab_noex;
        unsigned peek = solo ? pub[s] : pub[o];
        bool go = false;
        // hooks in the MFMA stream (k-block q of the section, a constant after unrolling):
        //   QF        the flag owed for the publish store of the section in front (drained by now)
        //   QX .. +7  one piece of the other set's input tile per block
        //   QP        look at the other set's flags (one load per lane)       QJ  judge
        //   QJ .. +7  one piece of the other set's gather per block
        constexpr int QF = 3, QX = 4;
        constexpr int QP = first ? 16 : 28, QJ = first ? 20 : 32;
        //   SOLO, all inside the input span:  QF as above   QX .. +7 the NEXT step's input tile   SP look at the own set's flags   SJ judge,
        //   SJ .. +7 the gather of h_{t-1}
#ifndef UP32_SQF
#define UP32_SQF 3
#endif
#ifndef UP32_SP
#define UP32_SP 14
#endif
#ifndef UP32_SJ
#define UP32_SJ 18
#endif
        constexpr int SP = UP32_SP, SJ = UP32_SJ;
        static_assert(!solo || (UP32_SQF < SP && SP + 2 <= SJ && SJ + NDMA <= KXB), "solo hook schedule");
        auto mid = [&](int q) {
            if (q == (solo ? UP32_SQF : QF)) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                raise_pending();
            }
            if constexpr (solo) {
                if (q >= QX && q < QX + NXD && !last && !ab_nox) issue_x_piece(bx ^ 1, tile, t + 1, q - QX);
                if constexpr (!first) {
                    if (q == SP) look_issue(look_lds, look_voff, fl_desc, fl_off + (unsigned)(s * NFL * sizeof(unsigned)));
                    if (q == SJ - 1) {
                        look_landed();
                        peek = *look_mine;
                    }
                    if (q == SJ) go = !ab_noex && __all((int)(peek >= pub[s])) != 0;
                    if (q >= SJ && q < SJ + NDMA && go) issue_h_piece(bh, s, pub[s], tile, t, q - SJ);
                }
                return;
            }
            if (q >= QX && q < QX + NXD && o_act) issue_x_piece(o, tile_of[o], step_of[o], q - QX);
            if (q == QP) look_issue(look_lds, look_voff, fl_desc, fl_off + (unsigned)(o * NFL * sizeof(unsigned)));     // (always: no branch)
            if (q == QJ - 1) {
                look_landed();
                peek = *look_mine;
            }
            if (q == QJ) go = o_h && __all((int)(peek >= pub[o])) != 0;
            if (q >= QJ && q < QJ + NDMA && go) issue_h_piece(o, o, pub[o], tile_of[o], step_of[o], q - QJ);
        };
        UP_STAMP(dg_top)
        // ---- stacked-gate product: one dependent chain of 32x32x2 MFMAs ---------------------------------------------------------
        f32x16 acc;
#pragma unroll
        for (int gate = 0; gate < 4; ++gate) {
            const f32x4 bv = bias_s[(wave * 4 + gate) * 2 + hh];
            acc[4 * gate] = bv[0]; acc[4 * gate + 1] = bv[1]; acc[4 * gate + 2] = bv[2]; acc[4 * gate + 3] = bv[3];
        }
        if (!ab_nomfma) {
            span32<KXB, NW>(acc, xb + bx * HL + frag, MR * 8, w, 0, [&](int q) { mid(q); });
            if constexpr (solo && !first) {
                // the slices h_{t-1}: on their way since block SJ, or -- a peer was late -- fetched now; one barrier and every wave sees them
                if (!go && !ab_noex) {
                    wait_flags(s, pub[s]);
#pragma unroll
                    for (int k = 0; k < NDMA; ++k) issue_h_piece(bh, s, pub[s], tile, t, k);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (!ab_nobar) bar();
            }
            if constexpr (!first) span32<BH, NW>(acc, hb + bh * HL + frag, MR * 8, w, 4 * KXB, [&](int q) { mid(KXB + q); });
        }
        // (a section shorter than the hook schedule -- step 0 of the narrow-input form is 4 k-blocks -- runs the rest of it here)
        {
            constexpr int NBLK = first ? KXB : KXB + BH;
#pragma unroll
            for (int q = NBLK; q < QJ + NDMA; ++q) mid(q);
        }
        // (the other set idle: THIS set runs the next section too, and its top copies into the buffers read above -- every wave
        //  must be through with them first.  `tile_of[o] < 0` is state, not a flag judgement: uniform over the workgroup, so the
        //  extra barrier pairs up.  With both sets active the copies into this set's buffers are issued from the other set's
        //  section, behind ITS top barrier.)
        if (!solo && tile_of[o] < 0 && !ab_nobar) bar();
        mfma_drain(acc);
        UP_STAMP(dg_chain)
        // ---- gates + cell update, lane-local: registers 4 gate + j = gate of unit 4 hh + j, window n ---------------------------
        float hnew[4];
        // (two cells at a time, the plain arithmetic on float2 values: one v_pk_* issue slot for both cells -- a wave's VALU work is serial
        //  with its MFMAs, tools/experiments/mfma_chain_rate.hip; same operations, same rounding as the scalar form)
#pragma unroll
        for (int j = 0; j < 4; j += 2) {
            if (ab_noact) {
                hnew[j] = acc[j] + acc[4 + j] + acc[8 + j] + acc[12 + j];
                hnew[j + 1] = acc[j + 1] + acc[5 + j] + acc[9 + j] + acc[13 + j];
                continue;
            }
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            auto exp2_2 = [](f32x2 v) { return f32x2{__builtin_amdgcn_exp2f(v[0]), __builtin_amdgcn_exp2f(v[1])}; };
            auto rcp_2 = [](f32x2 v) { return f32x2{__builtin_amdgcn_rcpf(v[0]), __builtin_amdgcn_rcpf(v[1])}; };
            const f32x2 ai = {acc[j], acc[j + 1]}, af = {acc[4 + j], acc[5 + j]}, ag = {acc[8 + j], acc[9 + j]}, ao = {acc[12 + j], acc[13 + j]};
            const f32x2 iv = rcp_2(1.0f + exp2_2(-1.4426950408889634f * ai));
            const f32x2 fv = rcp_2(1.0f + exp2_2(-1.4426950408889634f * af));
            const f32x2 gv = 2.0f * rcp_2(1.0f + exp2_2(-2.885390081777927f * ag)) - 1.0f;
            const f32x2 ov = rcp_2(1.0f + exp2_2(-1.4426950408889634f * ao));
            const f32x2 c = first ? iv * gv : fv * f32x2{cst[s][j], cst[s][j + 1]} + iv * gv;
            cst[s][j] = c[0]; cst[s][j + 1] = c[1];
            const f32x2 h = ov * (2.0f * rcp_2(1.0f + exp2_2(-2.885390081777927f * c)) - 1.0f);
            hnew[j] = h[0]; hnew[j + 1] = h[1];
        }
        if (abort_word != 0 || (solo && ctl[0] != 0)) return false;      // (a wave of this workgroup gave up in a blocking wait)
        if constexpr (solo) {
            prex[s] = !last && !ab_nox;
        } else {
            prex[o] = o_act;
            preh[o] = go;
        }
        if (!SEQ && last) {
            // ---- head: partial y over this wave's 8 units = four more MFMAs, the fresh h values are the activation fragment ----
            f32x16 ya;
#pragma unroll
            for (int i = 0; i < 16; ++i) ya[i] = 0.0f;
            const f32x4 wv = wo_s[wave * 64 + lane];
#pragma unroll
            for (int j = 0; j < 4; ++j) mfma32<false>(ya, wv[j], hnew[j]);
            mfma_drain(ya);
            // register 4 g + j of lane (n, hh) = target 8 g + 4 hh + j of window n; O <= 16: g = 0, 1
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
                *reinterpret_cast<f32x4*>(p.ypart + (((size_t)tile * MR + rn) * GH + member) * PO + 4 * oq) = sum;
            }
        }
        // ---- publish: this lane's four fresh h values are one 16-byte piece of the exchange layout; exactly ONE store
        //      instruction per wave and section ends it (the counted wait at the top of the next section relies on that) --------
        {
            const u32x4 hv = {__builtin_bit_cast(unsigned, hnew[0]), __builtin_bit_cast(unsigned, hnew[1]),
                              __builtin_bit_cast(unsigned, hnew[2]), __builtin_bit_cast(unsigned, hnew[3])};
            if constexpr (SEQ) {
                // every step's slices ARE the output: [tile][step][member][wave][window][8 units]; a last step's are awaited by
                // nobody inside the launch (its flag is raised all the same: one rule)
                const unsigned off = ab_noex ? 0x80000000u : (unsigned)(((size_t)tile * T + t) * SET_BYTES) + pub_off;
                if (in_l2) store_16<false>(hv, off, hx_desc);
                else store_16<true>(hv, off, hx_desc);
            } else {
                const unsigned off = (last || ab_noex) ? 0x80000000u : hx_base(s, (int)(pub[s] & 1u)) + pub_off;
                if (in_l2) store_16<false>(hv, off, hx_desc);
                else store_16<true>(hv, off, hx_desc);
            }
            if (!last && !ab_noex) {
                pub[s] += 1u;
                pend_idx = s * NFL + member * 4 + wave;
                pend_epoch = pub[s];
            }
        }
        UP_STAMP(dg_tail)
        // ---- next step / next tile of this set -------------------------------------------------------------------------------------
        if (last) {
            const int nt = tile + 2 * NC;
            tile_of[s] = nt < p.n_tiles ? nt : -1;
            step_of[s] = 0;
        } else {
            step_of[s] = t + 1;
        }
        return true;
    };

    bool ok = true;
    // (a cluster with ONE tile in all -- the launch has no more tiles than clusters -- runs the solo form, where the input span is long
    //  enough to hide the exchange: the 256-wide inputs)
    const bool solo_cluster = KXB >= 28 && tile_of[1] < 0 && tile_of[0] >= 0 && tile_of[0] + 2 * NC >= p.n_tiles && !(p.flags & APE_FLAG_ALT_FORM);
    if constexpr (KXB >= 28) {
        if (solo_cluster) {
#pragma unroll 1
            while (ok && tile_of[0] >= 0)
                ok = step_of[0] == 0 ? section(std::integral_constant<int, 0>{}, std::true_type{}, std::true_type{})
                                     : section(std::integral_constant<int, 0>{}, std::false_type{}, std::true_type{});
        }
    }
#pragma unroll 1
    while (ok && (tile_of[0] >= 0 || tile_of[1] >= 0)) {
        if (tile_of[0] >= 0)
            ok = step_of[0] == 0 ? section(std::integral_constant<int, 0>{}, std::true_type{}, std::false_type{})
                                 : section(std::integral_constant<int, 0>{}, std::false_type{}, std::false_type{});
        if (ok && tile_of[1] >= 0)
            ok = step_of[1] == 0 ? section(std::integral_constant<int, 1>{}, std::true_type{}, std::false_type{})
                                 : section(std::integral_constant<int, 1>{}, std::false_type{}, std::false_type{});
    }
    if (!ok) return;
    // (a flag still owed here is awaited by nobody -- the section behind a publish always raises it, and a set's last section
    //  publishes nothing -- so it is dropped: no flag store may be in flight when the last workgroup out zeroes the words)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

#ifdef APE_CLUSTER_STAMPS
    if (p.dbg_wg != nullptr && tid == 0 && cluster == 0 && member == 0) {
        p.dbg_wg[16] = dg_block_x; p.dbg_wg[17] = dg_block_h; p.dbg_wg[18] = dg_sections;
        p.dbg_wg[19] = dg_top; p.dbg_wg[20] = dg_chain; p.dbg_wg[21] = dg_tail;
    }
#endif
    // ---- self-cleaning: the last workgroup out re-zeroes every polled word ------------------------------------------
    __syncthreads();
    if (tid == 0)
        ctl[2] = (__hip_atomic_fetch_add(p.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) ? 1 : 0;
    __syncthreads();
    if (ctl[2] != 0) {
        const int n_flags = NC * 2 * NFL;
        for (int i = tid; i < n_flags; i += 256) __hip_atomic_store(p.xflags + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int i = tid; i < (int)gridDim.x; i += 256) __hip_atomic_store(xcc_words + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid < 8) __hip_atomic_store(class_ticket + tid * 16, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid == 0) __hip_atomic_store(p.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

constexpr size_t smem_upper() {
    return (size_t)4 * HL * sizeof(float) + (size_t)(4 * 4 * 2 + 4 * 64) * 16 + (size_t)4 * PO * MR * sizeof(float) + (size_t)4 * 64 * sizeof(unsigned) + 16;
}

// ---- the input of the layer-0 launch in fragment order: [tile][step][k-block 4][stream 32][8 features] ---------------------------
// One workgroup per (tile of 32 streams, step): ring order undone (step t lives in slot (t + x_ring) mod T), f64 z-score then
// float32 (estimator.py:103-104 + watch_phone_pocket_nn.py:100, the formula of lstm_tile16.hip), columns I .. 31 and streams
// past the bank zero.
__global__ __launch_bounds__(256) void ape_x_frag_kernel(const XFragParams q) {
    const unsigned tile = blockIdx.x / (unsigned)q.T, t = blockIdx.x - tile * (unsigned)q.T;
    const int slot = ((int)t + q.x_ring >= q.T) ? (int)t + q.x_ring - q.T : (int)t + q.x_ring;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int idx = threadIdx.x + 256 * e;                    // k-block idx / 256, stream (idx / 8) % 32, feature idx % 8
        const int kb = idx >> 8, row = (idx >> 3) & 31, k = kb * 8 + (idx & 7);
        const int stream = (int)tile * MR + row;
        float v = 0.0f;
        if (k < q.I && stream < q.S) {
            v = q.x[(size_t)stream * q.x_row_stride + (size_t)slot * q.I + k];
            if (q.xx_m != nullptr) v = (float)(((double)v - q.xx_m[k]) / q.xx_s[k]);
        }
        q.xfrag[((size_t)tile * q.T + t) * 1024 + idx] = v;
    }
}

// ---- a 256-wide input in fragment order: [tile][step][k-block 32][window 32][8] from the row-major [B * T][256] activations of
// ImuPoseLSTM's input layer (nn_models.py:242; row b * T + t).  One workgroup per (tile of 32 windows, step): the 32 rows are read
// whole (1 KiB each), turned through LDS, and leave as 1-KiB k-blocks; windows past the batch are zero.
constexpr int ZS = 8 * 32 + 8;           // LDS stride of a k-block (floats), as XS below

__global__ __launch_bounds__(256) void ape_z_frag_kernel(const float* __restrict__ z, float* __restrict__ zfrag, int B, int T) {
    __shared__ __attribute__((aligned(16))) float sl[32 * ZS];
    const unsigned tile = blockIdx.x / (unsigned)T, t = blockIdx.x - tile * (unsigned)T;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int idx = threadIdx.x + 256 * e;                    // float4 index: row idx / 64, columns 4 (idx % 64) ..
        const int row = idx >> 6, k = (idx & 63) * 4;
        const long long b = (long long)tile * 32 + row;
        f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
        if (b < B) v = *reinterpret_cast<const f32x4*>(z + ((size_t)b * T + t) * 256 + k);
        *reinterpret_cast<f32x4*>(sl + (k >> 3) * ZS + row * 8 + (k & 7)) = v;
    }
    __syncthreads();
    f32x4* dst = reinterpret_cast<f32x4*>(zfrag + ((size_t)tile * T + t) * 8192);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int idx = threadIdx.x + 256 * e;                    // float4 index: k-block idx / 64, inside it idx % 64
        dst[idx] = *reinterpret_cast<const f32x4*>(sl + (idx >> 6) * ZS + (idx & 63) * 4);
    }
}

// ---- the masked input of launch B, in the fragment order the kernel above copies ------------------------------------------------
// Sample row r (row_base + its index in this chunk) is sample r % n_mc of stream r / n_mc; its input at step t is that stream's
// layer-0 output h0[stream][t][unit] under the inter-layer dropout mask (nn.LSTM dropout, nn_models.py:169-174), drawn with the
// counters every fused kernel uses for its layer 0 (rows r & ~3, step, unit, layer; value index r & 3 -- lstm_tile16.hip), so
// the samples are the ones the batch-tile route draws.  One workgroup per (tile, step): thread = unit, one Philox call per four
// rows; the 32 KB go through LDS so that they leave as whole 1-KiB k-blocks [k-block][window 32][8 units].
constexpr int XS = 8 * MR + 8;           // LDS stride of a k-block (floats): the 8 k-blocks a wave writes side by side hit 8 bank groups

__global__ __launch_bounds__(256) void ape_mc_expand_kernel(const ExpandParams q) {
    __shared__ __attribute__((aligned(16))) float sl[BH * XS];
    const int unit = threadIdx.x;
    const unsigned tile = blockIdx.x / (unsigned)q.T, t = blockIdx.x - tile * (unsigned)q.T;
    const unsigned row0 = tile * MR;                              // first row of the tile, chunk-local
    const float keep = 1.0f / (1.0f - q.dropout_p);
    const bool drop = q.dropout_p > 0.0f;
    // stream of the tile's first row by ONE 32-bit division; the rows behind it count up (a 64-bit division per row, as the
    // first version had it, is ~100 scalar instructions each: the kernel was bound by them, 69 us for 25 600 rows)
    const unsigned g0 = (unsigned)q.row_base + row0;              // global index of the tile's first row (< 2^31)
    unsigned stream = g0 / (unsigned)q.n_mc, rem = g0 - stream * (unsigned)q.n_mc;
#pragma unroll
    for (int g = 0; g < MR / 4; ++g) {
        const unsigned r4 = g0 + 4 * g;                           // global index of the row quad (a multiple of 4)
        uint32_t rnd[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
        if (drop && q.masks == nullptr) philox4x32((uint32_t)r4, (uint32_t)t, (uint32_t)unit, (uint32_t)q.layer, (uint32_t)q.seed, (uint32_t)(q.seed >> 32), rnd);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float v = 0.0f;
            if (row0 + 4 * g + i < (unsigned)q.rows) {
                v = q.hseq_frag ? q.hseq[((size_t)(stream >> 5) * q.T + t) * HL + (unit >> 3) * (MR * 8) + (stream & 31) * 8 + (unit & 7)]
                                : q.hseq[((size_t)stream * q.T + t) * UH + unit];
                if (q.masks != nullptr) {          // the caller's multipliers (test hooks: the bank against the oracle under the same masks)
                    v *= q.masks[(((size_t)q.layer * q.masks_rows + (r4 + i)) * q.T + t) * UH + unit];
                } else if (drop) {
                    const float uf = (float)(rnd[i] >> 8) * (1.0f / 16777216.0f);
                    v = (uf >= q.dropout_p) ? v * keep : 0.0f;
                }
            }
            sl[(unit >> 3) * XS + (4 * g + i) * 8 + (unit & 7)] = v;
            if (++rem == (unsigned)q.n_mc) { rem = 0u; ++stream; }
        }
    }
    __syncthreads();
    f32x4* dst = reinterpret_cast<f32x4*>(q.xfrag + ((size_t)tile * q.T + t) * HL);
#pragma unroll
    for (int e = 0; e < HL / 4 / 256; ++e) {
        const int idx = threadIdx.x + 256 * e;                    // float4 index: k-block idx / 64, inside it idx % 64
        dst[idx] = *reinterpret_cast<const f32x4*>(sl + (idx >> 6) * XS + (idx & 63) * 4);
    }
}

// y[r][o] = b_out[o] + the eight members' partial sums, member 0 first (a fixed order: run-to-run identical bits)
__global__ __launch_bounds__(256) void ape_head_reduce_kernel(const float* __restrict__ ypart, const float* __restrict__ b_out,
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

bool ape_upper32_supported(int H, int L, int O) { return H == UH && L == 2 && O <= PO; }
size_t ape_upper32_xfrag_bytes(int rows, int T) { return (size_t)((rows + MR - 1) / MR) * T * SET_BYTES; }
size_t ape_lower32_xfrag_bytes(int streams, int T) { return (size_t)((streams + MR - 1) / MR) * T * 4 * 1024; }
size_t ape_lower32_hseq_bytes(int streams, int T) { return (size_t)((streams + MR - 1) / MR) * T * SET_BYTES; }
size_t ape_upper32_ypart_bytes(int rows) { return (size_t)((rows + MR - 1) / MR) * MR * GH * PO * sizeof(float); }

hipError_t ape_prepare_lstm_upper32() {
    static_assert(smem_upper() <= APE_LDS_BYTES, "LDS layout exceeds a CU");
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ape_lstm_upper32<32, false>), hipFuncAttributeMaxDynamicSharedMemorySize, APE_LDS_BYTES);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ape_lstm_upper32<4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, APE_LDS_BYTES);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&ape_lstm_upper32<32, true>), hipFuncAttributeMaxDynamicSharedMemorySize, APE_LDS_BYTES);
}

// A two-layer 256-unit LSTM with a 256-wide input (ImuPoseLSTM behind its input layer), one layer per launch on the persistent clusters:
// (1) the row-major activations -> fragment order; (2) layer 0 in the SEQ form with the wide input (<32, true>): every step's slices go to
// p0.hseq in exactly the order (3) layer 1 (<32, false>) copies its input tiles from -- no builder in between (eval mode: no mask); head
// partials + reduce as in the bank.  p1.xfrag must be p0.hseq.
hipError_t ape_launch_lstm_split32(const float* z, int B, const UpperParams& p0, const UpperParams& p1, const float* b_out, float* y,
                                   int max_clusters, hipStream_t stream) {
    if (p0.n_tiles < 1 || p0.n_tiles != p1.n_tiles || max_clusters < 8 || p1.O > PO || p1.xfrag != p0.hseq) return hipErrorInvalidValue;
    hipLaunchKernelGGL(ape_z_frag_kernel, dim3(p0.n_tiles * p0.T), dim3(256), 0, stream, z, const_cast<float*>(p0.xfrag), B, p0.T);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    int clusters = (p0.n_tiles + 7) / 8 * 8;
    if (clusters > max_clusters) clusters = max_clusters;
    hipLaunchKernelGGL((ape_lstm_upper32<32, true>), dim3(clusters * GH), dim3(256), smem_upper(), stream, p0);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((ape_lstm_upper32<32, false>), dim3(clusters * GH), dim3(256), smem_upper(), stream, p1);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(ape_head_reduce_kernel, dim3((B * PO + 255) / 256), dim3(256), 0, stream, p1.ypart, b_out, y, B, p1.O);
    return hipGetLastError();
}

// layer 0 of the S streams of a Monte-Carlo bank, every step's output in fragment order (p.hseq): the input tiles first
// (ape_x_frag_kernel: ring order undone, f64 z-score, zero padding to 32 columns), then the SEQ form of the cluster kernel
hipError_t ape_launch_lstm_lower32(const UpperParams& p, const XFragParams& xq, int max_clusters, hipStream_t stream) {
    if (p.n_tiles < 1 || max_clusters < 8) return hipErrorInvalidValue;
    hipLaunchKernelGGL(ape_x_frag_kernel, dim3(p.n_tiles * p.T), dim3(256), 0, stream, xq);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    int clusters = (p.n_tiles + 7) / 8 * 8;
    if (clusters > max_clusters) clusters = max_clusters;
    hipLaunchKernelGGL((ape_lstm_upper32<4, true>), dim3(clusters * GH), dim3(256), smem_upper(), stream, p);
    return hipGetLastError();
}

// one chunk of sample rows: expand -> upper layer -> head reduce, all on `stream`.  `max_clusters` = 32-row clusters the device
// holds at once (a multiple of 8: whole block-index classes); the grid is the smaller of that and the tiles, rounded up to 8.
hipError_t ape_launch_lstm_upper32(const UpperParams& p, const ExpandParams& q, const float* b_out, float* y, int max_clusters,
                                   hipStream_t stream, hipEvent_t ev_begin, hipEvent_t ev_end) {
    if (p.n_tiles < 1 || max_clusters < 8 || p.O > PO) return hipErrorInvalidValue;
    hipLaunchKernelGGL(ape_mc_expand_kernel, dim3(p.n_tiles * p.T), dim3(256), 0, stream, q);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    int clusters = (p.n_tiles + 7) / 8 * 8;
    if (clusters > max_clusters) clusters = max_clusters;
    if (ev_begin) (void)hipEventRecord(ev_begin, stream);          // (measurement aid: ape_streams_profile)
    hipLaunchKernelGGL((ape_lstm_upper32<32, false>), dim3(clusters * GH), dim3(256), smem_upper(), stream, p);
    e = hipGetLastError();
    if (ev_end) (void)hipEventRecord(ev_end, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(ape_head_reduce_kernel, dim3((q.rows * PO + 255) / 256), dim3(256), 0, stream, p.ypart, b_out, y, q.rows, p.O);
    return hipGetLastError();
}
