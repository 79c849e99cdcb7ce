// Weight-stationary "cluster" LSTM kernel for gfx950 (MI355X).
//
// Same arithmetic as lstm_tile16.hip (reference estimate/nn_models.py:169-174,180-189: L stacked
// LSTM layers, gates i,f,g,o, zero initial state, Linear head on the last step) but mapped so that
// the 3.3 MB of weights never move after the prologue:
//
//   * GH = H/16 workgroups (one per CU) form a CLUSTER that owns MR = 16*NMT windows for all T
//     steps.  Member m owns hidden units [16m, 16m+16) of EVERY layer with all four gates; wave w
//     of it owns 4 of those units = one 16-column MFMA tile (column = unit*4 + gate).
//   * each wave keeps its slice of [W_ih | W_hh] of every layer in REGISTERS for the whole launch
//     (AGPRs, already in v_mfma_f32_16x16x4_f32 fragment order: 200 per lane for the pocket model), so
//     the weight operand of the MFMA needs no load at all.  At B=1024 all 256 CUs are busy
//     (16 clusters x 16 CUs) instead of 64, and no CU re-streams weights every step.
//   * the activation operand (h of all H units of the cluster's windows, and x_t) lives in LDS; every
//     layer-step each member publishes its 16-unit slice of h to a small exchange buffer and
//     gathers the other members' slices.  Hand-off protocol (placement-independent, MI355X guide
//     G16 / visibility table row 1): payload by 16-byte sc1 (write-through) buffer stores, each
//     wave stores the pieces of its own four units, drains vmcnt(0) and then raises ITS epoch flag
//     with an agent-scope relaxed atomic; every consumer wave polls all 4*GH wave flags of a layer with
//     one agent-scope relaxed load per lane and only then issues loads of the payload, every one a
//     16-byte sc1 buffer load.  Exchange buffers are double-buffered by step parity; the last
//     workgroup to finish re-zeroes the flags and counters (self-cleaning, no memset node); every spin
//     is bounded and raises a status word.
//   * everything but the gate math rides BETWEEN MFMAs (hook() in layer_mfma, one call per 16-MFMA
//     k-block): the flag owed for the previous section's slice, the look at the next layer's flags, the
//     gather loads, the LDS commits of gathered slices, the f64 z-score of the next x row.  A phase has one
//     barrier per layer above 0 (in front of the first read of its recurrent buffer) and one at its end.
//   * the layers are software-pipelined: in phase p layer l works on step t = p - l, so all the
//     layer computations of a phase depend only on the previous phase and the exchange of one
//     layer's slice flies under the other layers' MFMAs (see the k-block schedule at the phase loop).
//   * the MFMA takes the weights as its A operand and the activations as B, with the wave's 16 gate
//     columns ordered unit*4 + gate: every lane then holds i,f,g,o of ONE (unit, batch row) in its four
//     accumulator registers, so activations and the c/h update are lane-local and spread over all 64 lanes.
#include "ape_internal.h"
#include "async_look.h"
#include "../../include/ape_hip.h"

// k-block schedule of the exchange work under the MFMAs (see the phase loop); tunable at build time
#ifndef APE_QF
#define APE_QF 1      // block in which the flag owed from the last section goes up
#endif
#ifndef APE_QP
#define APE_QP 6      // block in which the next layer's flags are looked at
#endif
#ifndef APE_QJ
#define APE_QJ 3      // blocks between the look and the verdict (= first gather block)
#endif
#ifndef APE_PPB
#define APE_PPB 8     // gather pieces put into flight per block
#endif
#ifndef APE_CPB
#define APE_CPB 4     // gathered pieces committed to LDS per block
#endif

namespace {

#include "lstm_cluster_common.h"

template <int H, int L, int KX, int NMT, bool DROP>
__global__ __launch_bounds__(256, 1) void ape_lstm_cluster(const ClusterParams p) {
    constexpr int GH = H / 16;            // workgroups (CUs) per cluster
    constexpr int MR = 16 * NMT;          // windows per cluster
    constexpr int SH = H + 8;             // LDS row strides (floats): conflict-free ds_read_b128
    constexpr int SX = KX + 8;
    constexpr int SO = 20;                // own-slice staging row stride
    constexpr int QX = KX / 16, QH = H / 16;
    constexpr int NW0 = (KX + H) / 4;     // weight registers per lane, layer 0
    constexpr bool ACC_V = NW0 + (L - 1) * (2 * H / 4) + 4 * NMT > 256;    // weights fill the accumulator file: accumulators in VGPRs
    constexpr int NW1 = (2 * H) / 4;      //                            layers >= 1
    constexpr int TPS = 4 * MR;           // threads that move one member slice (16 B each)
    constexpr int SPP = 256 / TPS;        // slices per gather pass
    // inter-layer dropout (train-mode nn.LSTM after monte_carlo_predictions, nn_models.py:204): the slice of a
    // layer below the top is published twice -- raw (its own recurrence) and masked (the next layer's input)
    constexpr int NV = DROP ? 2 : 1;
    static_assert(!DROP || 2 * TPS <= 256, "dropout variants are built for NMT <= 2");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    // D layout: lane (r = lane&15, g = lane>>4) holds gates i,f,g,o (register 0..3) of unit g for batch row r
    // Cluster membership by ARRIVAL TICKET, not by blockIdx: the first GH workgroups to start form cluster 0,
    // the next GH cluster 1, ...  Every member of a formed cluster is therefore resident, so formed clusters
    // always make progress and free their CUs on exit; the launch cannot deadlock as long as one cluster fits
    // on the chip, whatever the dispatch order or however many CUs other work occupies.
    const int T = p.T, I = p.I, O = p.O;
    const bool normalize = (p.flags & APE_FLAG_NORMALIZE_INPUT) != 0;
    const bool bcast_x = (p.flags & APE_FLAG_BROADCAST_X) != 0;   // monte_carlo_predictions: one window, B rows
#ifdef APE_CLUSTER_STAMPS
    const bool diag_noex = (p.flags & APE_DIAG_NO_EXCHANGE) != 0;    // timing only: skip polls/gathers/publishes
    const bool diag_noact = (p.flags & APE_DIAG_NO_ACT) != 0;        // timing only: skip the transcendental math
#else
    constexpr bool diag_noex = false, diag_noact = false;            // the shipped library has no diagnostic paths
#endif
    const bool drop_masks = DROP && (p.flags & APE_FLAG_DROPOUT_MASKS) != 0;
    // diagnostic stamps (APE_DIAG_STAMP): shader clock vs the 100 MHz real-time counter around the phase
    // loop; written to words 4..7 behind the status word, which no other code reads
    unsigned long long stamp_c0 = 0, stamp_r0 = 0;
    if (p.flags & APE_DIAG_STAMP) { stamp_c0 = __builtin_amdgcn_s_memtime(); stamp_r0 = __builtin_amdgcn_s_memrealtime(); }
#ifdef APE_CLUSTER_STAMPS
    const unsigned long long wg_t0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long wg_t1 = 0, wg_t2 = 0;
#endif

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* hbuf = smem;                           // [L][MR][SH]  gathered h of every layer
    float* xin = hbuf + L * MR * SH;              // [MR][SX]
    float* own = xin + MR * SX;                   // [NV][MR][SO] this member's fresh slice (raw, masked)
    float* dbuf = own + NV * MR * SO;             // [L-1][MR][SH] gathered MASKED h (DROP only)
    unsigned* look_s = reinterpret_cast<unsigned*>(dbuf + (DROP ? (L - 1) * MR * SH : 0));   // [wave 4][64]: landing zones of the flag looks (async_look.h)
    int* ctl = reinterpret_cast<int*>(look_s + 4 * 64);         // [0] abort flag, [1] arrival ticket, [2] last-out
    unsigned* arrive = reinterpret_cast<unsigned*>(ctl + 4);    // [L]: publishes of layer l whose stores have drained, over this member's four waves
    // A launch that finds the sticky status word set -- an earlier launch on this model aborted and skipped its
    // self-cleaning, so tickets, flags and counters are stale -- leaves without touching anything (status 2 tells
    // ape_model_check that later launches ran into it), and so does a workgroup whose ticket lies outside the grid.
    // XCD classes (round 3; the host sets APE_FLAG_XCD_CLASSES when the grid is whole groups of 8 clusters): workgroups of one block-index
    // class (blockIdx % 8 -- round-robin over the 8 XCDs) form their clusters among themselves, so a cluster's members share an XCD and
    // its L2; verified at run time below (XCC_ID of every member): then the slices are handed over by PLAIN stores that stay in that L2
    // instead of write-through ones -- the form lstm_cluster32.hip introduced, here for every launch of the first generation.
    const bool cls_mode = (p.flags & APE_FLAG_XCD_CLASSES) != 0;
    unsigned* const class_ticket = p.xcc_slots + 64;
    unsigned* const xcc_words = p.xcc_slots + 64 + 8 * 16;
    const int cls = blockIdx.x & 7;
    if (threadIdx.x < L) arrive[threadIdx.x] = 0u;
    if (threadIdx.x == 0) {
        ctl[0] = 0;
        ctl[1] = -1;
        ctl[3] = 0;
        if (__hip_atomic_load(p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
            const unsigned tk = cls_mode ? __hip_atomic_fetch_add(class_ticket + cls * 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                         : __hip_atomic_fetch_add(p.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (tk < (cls_mode ? gridDim.x / 8 : gridDim.x)) ctl[1] = (int)tk;
            else __hip_atomic_store(p.status, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    if (ctl[1] < 0) return;
    const int ticket = __builtin_amdgcn_readfirstlane(ctl[1]);
    const int cluster = cls_mode ? (ticket / GH) * 8 + cls : ticket / GH, member = ticket % GH;
    const int row0 = cluster * MR;
    unsigned my_xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(my_xcc));
    my_xcc &= 0xFu;
    if (cls_mode && threadIdx.x == 0)
        __hip_atomic_store(xcc_words + cluster * GH + member, 0x10u | my_xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    f32x4 bias_r[L];
#pragma unroll
    for (int l = 0; l < L; ++l)
#pragma unroll
        for (int k = 0; k < 4; ++k) bias_r[l][k] = p.bias[l][k * H + member * 16 + wave * 4 + g];

    float cst[L][NMT];                // cell state of (unit g, batch row 16*mt + r)
#pragma unroll
    for (int l = 0; l < L; ++l)
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt) cst[l][mt] = 0.0f;

    // exchange buffer descriptors (wave-uniform: kernel arguments only)
    const __amdgpu_buffer_rsrc_t hx_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(p.hx, 0, (int)p.hx_bytes, 0x00020000);
    constexpr int NFL = 4 * GH;                  // flags per (cluster, layer): one per member WAVE
    const ape_desc_t fl_desc = ape_make_desc(p.xflags, (unsigned)((gridDim.x / GH) * L * NFL * sizeof(unsigned)));
    const unsigned fl_off = (unsigned)(cluster * L * NFL * sizeof(unsigned));
    const unsigned look_voff = (unsigned)((lane & (NFL - 1)) * sizeof(unsigned));
    const unsigned look_lds = (unsigned)reinterpret_cast<unsigned long long>(look_s) + (unsigned)(wave * 256);
    const unsigned* const look_mine = look_s + wave * 64 + lane;
    unsigned* const myflags = p.xflags + (size_t)cluster * L * NFL;

    // ---- exchange helpers ------------------------------------------------------------------------------
    constexpr int NGV = GH / SPP;                    // 16-byte pieces each thread moves per gather
    const int g_sl = tid / TPS, g_idx = tid - g_sl * TPS;       // slice within a pass, piece within the slice
    const int g_row = g_idx % MR, g_quad = g_idx / MR;           // row fastest: a wave's 64 pieces are 1 KiB contiguous

    // every wave polls for itself (no barrier on the way): all members published epoch `want` of layer l?
    // bounded; on give-up raises the sticky status word and the workgroup abort flag
    // non-blocking look: the load only.  Eval-mode instantiations issue it as LDS-DMA into the wave's landing zone (async_look.h; rounds 3-4:
    // an asm load with a register destination, which hipcc is free to copy or re-use while the load is in flight) and read it back at the judge
    // (peek_wait): hipcc hoists the comparison of a compiler-visible load up to the load and waits `vmcnt(0)` right behind it, an
    // L2 round trip exposed in every section (round 3, found in the disassembly; lstm_cluster32.hip).  Measured on this kernel:
    // upper-arm model 1024 x 64 530 -> 513 us, ImuPoseLSTM neutral -- but the dropout instantiations (the estimators' 25-sample
    // Monte-Carlo launch, T = 6) LOSE 1 us of 43 with it (there the exposed wait gives late flags time to arrive), so they keep
    // the compiler-visible form.
    auto peek_flags = [&](int l, unsigned want) -> unsigned {
        if constexpr (DROP) {
            if (diag_noex || lane >= NFL) return want;
            return __hip_atomic_load(myflags + l * NFL + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            look_issue(look_lds, look_voff, fl_desc, fl_off + (unsigned)(l * NFL * sizeof(unsigned)));
            return 0u;                                   // (the value comes out of peek_wait)
        }
    };
    auto peek_wait = [&](unsigned& v) {
        if constexpr (!DROP) {
            look_landed();
            v = *look_mine;
        }
    };
    auto wait_flags = [&](int l, unsigned want, unsigned peeked) {
        if (diag_noex) return;
        if (__all((int)(peeked >= want))) return;                  // the prefetched look was enough
        unsigned spins = 0;
        while (true) {
            unsigned v = want;
            if (lane < NFL)
                v = __hip_atomic_load(myflags + l * NFL + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__all((int)(v >= want))) return;
            if (++spins > SPIN_LIMIT ||
                __hip_atomic_load(p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                if (lane == 0) {
                    ctl[0] = 1;
                    __hip_atomic_store(p.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                return;
            }
            __builtin_amdgcn_s_sleep(2);
        }
    };
    // EVERY load of handed-off bytes is a 16-byte sc1 buffer load (bypasses this CU's L1)
    // the NGV pieces move in two halves of NGH so that at most NGH*4 registers hold in-flight slices
    constexpr int NGH = NGV * NV;     // whole gather in flight at once (the weights live in AGPRs, VGPRs are free)
    constexpr unsigned SLICE_SET = GH * MR * 16 * sizeof(float);       // bytes of one (layer, parity, variant)
    auto hx_base = [&](int l, int par, int v) -> unsigned {
        return (unsigned)(((((size_t)cluster * L + l) * 2 + par) * NV + v) * SLICE_SET);
    };
    // piece kk of the gather of layer l: variant kk / NGV (0 raw, 1 masked), pass kk % NGV.  `goff` is the thread's
    // byte offset inside a pass, or that plus 2^31 -- outside the descriptor, so the load returns zeros without
    // touching memory -- while the flags have not been seen raised (the uniform part sits in soffset, which the
    // range check ignores)
    // exchange layout [member][wave quad][row][4 units]: what ONE wave stores per layer-step (64 rows x 16 B) is one
    // contiguous KiB -- whole 64-byte lines, not a quarter of each row's line
    const unsigned g_thread_off = (unsigned)((((g_sl * 4 + g_quad) * MR + g_row) * 4) * sizeof(float));
    auto issue_piece = [&](int l, int par, int kk, unsigned goff, f32x4 (&gv)[NGH]) {
        if (diag_noex) return;
        const int v = kk / NGV, k = kk - v * NGV;
        if (v == 1 && l == L - 1) return;                // the top layer has no masked variant
        gv[kk] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
            hx_rsrc, goff, hx_base(l, par, v) + (unsigned)(k * SPP * MR * 16 * sizeof(float)), 16 /* sc1 */));
    };
    auto issue_gather = [&](int l, int par, f32x4 (&gv)[NGH]) {
#pragma unroll
        for (int kk = 0; kk < NGH; ++kk) issue_piece(l, par, kk, g_thread_off, gv);
    };
    auto commit_gather = [&](int l, const f32x4 (&gv)[NGH]) {
        if (diag_noex) return;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            if (v == 1 && l == L - 1) break;
            float* dst = (v == 0) ? hbuf + l * MR * SH : dbuf + l * MR * SH;
#pragma unroll
            for (int k = 0; k < NGV; ++k) {
                const int m = k * SPP + g_sl;
                *reinterpret_cast<f32x4*>(dst + g_row * SH + m * 16 + 4 * g_quad) = gv[v * NGV + k];
            }
        }
    };
    auto commit_piece = [&](int l, int kk, const f32x4 (&gv)[NGH]) {
        if (diag_noex) return;
        const int v = kk / NGV, k = kk - v * NGV;
        if (v == 1 && l == L - 1) return;
        float* dst = (v == 0) ? hbuf + l * MR * SH : dbuf + l * MR * SH;
        *reinterpret_cast<f32x4*>(dst + g_row * SH + (k * SPP + g_sl) * 16 + 4 * g_quad) = gv[kk];
    };
    // workgroup barrier that waits for this wave's LDS traffic only (not for loads/stores in flight to memory)
    auto bar = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    // blocking form (pipeline fill/drain and the head): wait, load, barrier (no reader of the target left), commit,
    // barrier
    auto gather_now = [&](int l, unsigned want, int par) -> bool {
        f32x4 ga[NGH];
        wait_flags(l, want, 0u);
        issue_gather(l, par, ga);
        bar();
        commit_gather(l, ga);
        bar();
        return ctl[0] == 0;
    };

    // ---- x staging: thread owns NE (row, k) elements of the [MR][KX] step slab, all with the same k ----
    // Loads go through a buffer descriptor over this cluster's rows: rows past the batch and the padded columns
    // k >= I fall outside it and read as 0 (no predicates, 32-bit offsets).
    constexpr int NE = (MR * KX) / 256;
    constexpr int RPE = 256 / KX;                 // rows between a thread's consecutive elements
    const int xk = tid % KX, xrow = tid / KX;
    const int rows_here = bcast_x ? MR : max(0, min(MR, p.B - row0));      // (whole clusters past the batch: class-mode grids are rounded up)
    // (xrs: floats between rows -- T * I, or the caller's stride when the rows are the first window slots of a Monte-Carlo bank's ring;
    //  a row's T * I floats are all that is ever addressed, the descriptor's length only has to be no shorter)
    const unsigned xrs = p.x_row_stride != 0 ? (unsigned)p.x_row_stride : (unsigned)(T * I);
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x) + (bcast_x ? (size_t)0 : (size_t)row0 * xrs), 0,
        (int)(bcast_x ? (size_t)T * I * sizeof(float) : ((size_t)(rows_here > 0 ? rows_here - 1 : 0) * xrs + (rows_here > 0 ? (size_t)T * I : 0)) * sizeof(float)), 0x00020000);
    const unsigned x_off0 = (xk < I) ? (unsigned)(((bcast_x ? 0u : (unsigned)xrow * xrs) + (unsigned)xk) * sizeof(float)) : 0x80000000u;
    const unsigned x_estride = bcast_x ? 0u : (unsigned)(RPE * xrs * sizeof(float));
    float xr[NE];
    auto fetch_x = [&](int t) {                    // t < T (the step offset is not range-checked)
        const int slot = (t + p.x_ring >= T) ? t + p.x_ring - T : t + p.x_ring;      // windows kept as rings
#pragma unroll
        for (int e = 0; e < NE; ++e)
            xr[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                        x_rsrc, x_off0 + (unsigned)e * x_estride, (unsigned)(slot * I * sizeof(float)), 0));
    };
    // f64 z-score, cast f32 (estimator.py:103-104, watch_phone_pocket_nn.py:100), into LDS.  Per-thread constants
    // (every element a thread stages has the same k): mean, std and the host-rounded 1/std; (0, 1, 1) passes x
    // through exactly when the caller normalised already.
    const double x_mean = (normalize && xk < I) ? p.xx_m[xk] : 0.0;
    const double x_std = (normalize && xk < I) ? p.xx_s[xk] : 1.0;
    const double x_rstd = (normalize && xk < I) ? p.xx_r[xk] : 1.0;
    auto stage_elem = [&](int e) {
        // (x - m) / s in f64, correctly rounded: q0 = d * (1/s), one residual step q0 + (d - q0 s)(1/s)
        // (the tail of the hardware division sequence; a NaN residual means s = 0 or inf -> keep q0,
        // which is then the same +-inf / NaN the division yields)
        const double d = (double)xr[e] - x_mean;
        const double q0 = d * x_rstd;
        const double rr = fma(-q0, x_std, d);
        const double q1 = fma(rr, x_rstd, q0);
        xin[(xrow + e * RPE) * SX + xk] = (float)((rr == rr) ? q1 : q0);
    };
    auto stage_x = [&]() {
#pragma unroll
        for (int e = 0; e < NE; ++e) stage_elem(e);
    };
    fetch_x(0);
    // ---- weights: registers, for the whole launch.  Issued AFTER the bias and x_0 loads (loads return in order): the
    //      first phase needs x_0, the bias and only the 8 x-span registers of layer 0, so the other ~190 loads per lane
    //      land under phase 0 and the pipeline-fill exchange instead of in front of them
    static_assert(NW0 % 4 == 0 && NW1 % 4 == 0, "weight registers are loaded four at a time");
    float w0[NW0];
    float w1[L > 1 ? NW1 : 1];
    float w2[L > 2 ? NW1 : 1];
    {
        // 16 bytes per lane and load (host order [k-quad][lane][4]): a quarter of the instructions of a dword walk
        const f32x4* s0 = reinterpret_cast<const f32x4*>(p.wcl[0]) + ((size_t)(member * 4 + wave) * (NW0 / 4)) * 64 + lane;
#pragma unroll
        for (int i = 0; i < NW0 / 4; ++i) {
            const f32x4 v = s0[i * 64];
            w0[4 * i] = v[0]; w0[4 * i + 1] = v[1]; w0[4 * i + 2] = v[2]; w0[4 * i + 3] = v[3];
        }
        if constexpr (L > 1) {
            const f32x4* s1 = reinterpret_cast<const f32x4*>(p.wcl[1]) + ((size_t)(member * 4 + wave) * (NW1 / 4)) * 64 + lane;
#pragma unroll
            for (int i = 0; i < NW1 / 4; ++i) {
                const f32x4 v = s1[i * 64];
                w1[4 * i] = v[0]; w1[4 * i + 1] = v[1]; w1[4 * i + 2] = v[2]; w1[4 * i + 3] = v[3];
            }
        }
        if constexpr (L > 2) {
            const f32x4* s2 = reinterpret_cast<const f32x4*>(p.wcl[2]) + ((size_t)(member * 4 + wave) * (NW1 / 4)) * 64 + lane;
#pragma unroll
            for (int i = 0; i < NW1 / 4; ++i) {
                const f32x4 v = s2[i * 64];
                w2[4 * i] = v[0]; w2[4 * i + 1] = v[1]; w2[4 * i + 2] = v[2]; w2[4 * i + 3] = v[3];
            }
        }
    }
    stage_x();
    if (T > 1) fetch_x(1);
    // ---- class mode: do all members of this cluster really share an XCD?  (behind the weight loads: the peers arrive meanwhile) ------
    if (cls_mode && wave == 0) {
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
    bar();
    if (ctl[0] != 0) return;
        // hand-over form (ape_internal.h): write-through (`sc1`) payload stores unless the caller opted into the plain in-XCD form AND the
    // members were verified to share an XCD; uniform over the cluster
    const bool in_l2 = APE_HANDOVER_IN_L2(p.flags, ctl[3] != 0);

    // The flag a wave owes for the slice it stored last: raised once those stores have drained -- a few k-blocks
    // into the NEXT section's MFMAs (the write-through latency hides there), or at the latest before this wave
    // blocks on anybody else's flag.
    // Raised per MEMBER (round 6): every wave arrives on the layer's LDS counter once its own stores have drained, and the last of the four
    // stores the member's four flag words in ONE instruction.  32 .. 64 waves each storing its own word of the same one or two cache lines
    // at about the same time complete one after the other on the memory side; the last of them became visible microseconds after it was
    // issued (measured on lstm_cluster16.hip's final gather: 3 .. 7 us, profiles/r06_flag_serialisation.md).
    int pend_idx = -1;                      // layer of the pending publish
    unsigned pend_epoch = 0u;
    auto raise_pending = [&]() {
        if (pend_idx < 0) return;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this wave's slice stores are complete
        unsigned prev = 0u;
        if (lane == 0) prev = __hip_atomic_fetch_add(arrive + pend_idx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        prev = __builtin_amdgcn_readfirstlane(prev);
        if (!diag_noex && prev + 1u == 4u * pend_epoch && lane < 4)
            __hip_atomic_store(myflags + pend_idx * NFL + member * 4 + lane, pend_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pend_idx = -1;
    };

#ifdef APE_CLUSTER_STAMPS
    wg_t1 = __builtin_amdgcn_s_memrealtime();
#endif
    STAMP_DECL
    const int P = T + L - 1;
    // Section (ph, l) = layer l on step t = ph - l.  What it gathers is what the NEXT section in program order
    // needs: the freshly published slices of layer ln.  Where the gathered registers go:
    //   * ln >= 1: they are CARRIED into section ln and committed to hbuf[ln] during its input span (which reads
    //     hbuf[ln-1] only); the mid-section barrier in front of the recurrent span makes them visible;
    //   * ln == 0 (gathered by the top layer's section): committed to hbuf[0] in the last k-blocks of that same
    //     section, after its mid-section barrier (no reader of hbuf[0] is left), visible by its end barrier.
    // So a phase has one barrier per layer above 0 plus one at its end, none of them waiting on memory, and every
    // LDS commit, flag and gather instruction sits between MFMAs.
    bool prefetched = false;          // the slices the NEXT section reads are in hbuf or on their way there in gv
    bool carry = false;               // gv holds slices the next section has to commit to its recurrent buffer
    f32x4 gv[NGH];
#pragma unroll 1
    for (int ph = 0; ph < P; ++ph) {
#pragma unroll
        for (int l = 0; l < L; ++l) {
            const int t = ph - l;                      // the step layer l works on in this phase
            // h^l_{t-1} (published one phase ago) feeds layer l at step t AND layer l+1 at step t-1, so it is
            // gathered whenever it exists, also in the pipeline tail where layer l itself is done
            const bool have_prev = (t >= 1 && t <= T) && !(l == L - 1 && t == T);
            const bool active = (t >= 0 && t < T);      // both uniform over the whole grid
            STAMP_BEGIN();
            if (have_prev && !prefetched) {             // pipeline fill / drain only
                raise_pending();
                if (!gather_now(l, (unsigned)t, (t - 1) & 1)) return;
            }
            prefetched = false;
            if (!active) {
                if (carry) {                            // (tail) nobody will run the input span that commits them
                    raise_pending();
                    bar();
                    commit_gather(l, gv);
                    bar();
                    carry = false;
                }
                continue;
            }
            STAMP_END(1);                                // 1: blocking gathers (pipeline fill / drain only)

            // next section in program order: (ph, l+1) or (ph+1, 0)
            const int ln = (l + 1 < L) ? l + 1 : 0;
            const int tn = (l + 1 < L) ? t - 1 : t + L;          // = ph_n - l_n
            const bool pre = (tn >= 1 && tn <= T) && !(ln == L - 1 && tn == T) && (l + 1 < L || ph + 1 < P);
            const bool last = (l == L - 1);
            const bool carry_in = carry;
            carry = false;
            // k-block schedule of the work under the MFMAs (one call of `hook` per k-block):
            //   0 .. CBLK-1   (l >= 1) commit the carried slices, CPB pieces per block
            //   QF            raise the flag owed for the slice stored at the end of the last section
            //   QP / Q0       look at / judge the flags of layer ln; from Q0 on PPB gather pieces per block -- to real
            //                 addresses if every flag was up, else outside the descriptor (zeros, no traffic) and the
            //                 blocking path behind the MFMAs redoes them
            //   QM = QIN-1    (l >= 1) mid-section barrier, in front of the first read of the recurrent buffer
            //   QIN ..        (l == 1) f64 z-score of x_{ph+1} into LDS, then the fetch of x_{ph+2}
            //   QC ..         (top layer) commit the slices just gathered to hbuf[0]
            constexpr int PPB = APE_PPB, GBLK = (NGH + PPB - 1) / PPB;
            constexpr int CPB = APE_CPB, CBLK = (NGH + CPB - 1) / CPB;
            constexpr int EPB = (NE + QH - 1) / QH, XBLK = (NE + EPB - 1) / EPB;
            constexpr int QF = APE_QF;
            const int nblk = (l == 0) ? QX + QH : 2 * QH;
            const int QIN = (l == 0) ? QX : QH;
            const int QP = (nblk >= 16) ? APE_QP : APE_QP - 1;
            const int Q0 = QP + APE_QJ;
            const int QM = QIN - 1;
            const int QC = nblk - CBLK;
            static_assert(CBLK <= QH - 1 && APE_QP + APE_QJ + GBLK <= 2 * QH - CBLK && CBLK <= APE_QP + APE_QJ - 1 + 1,
                          "k-block schedule does not fit");
            unsigned peeked = 0u;
            bool peek_pending = false;                   // (eval-mode form) a look is in flight and has not been waited for
            bool ready = false, new_done = false;
            unsigned goff = g_thread_off + 0x80000000u;
            auto hook = [&](int q) {
                if (l >= 1 && q < CBLK && carry_in) {
#pragma unroll
                    for (int j = 0; j < CPB; ++j)
                        if (q * CPB + j < NGH) commit_piece(l, q * CPB + j, gv);
                }
                if (q == QF) raise_pending();
                // (L == 1, the layer-0 form of a Monte-Carlo bank's launch A: the slices the next section reads are the ones THIS section
                //  publishes at its end -- nothing to look at or gather under its MFMAs; the blocking exchange sits behind the publish)
                if (L > 1 && q == QP) { peeked = peek_flags(ln, (unsigned)tn); peek_pending = true; }
                if (L > 1 && q == Q0) {
                    if (peek_pending) { peek_wait(peeked); peek_pending = false; }
                    ready = pre && (diag_noex || __all((int)(peeked >= (unsigned)tn)) != 0);
                    goff = g_thread_off + (ready ? 0u : 0x80000000u);
                }
                if (L > 1 && q >= Q0 && q < Q0 + GBLK) {
#pragma unroll
                    for (int j = 0; j < PPB; ++j)
                        if ((q - Q0) * PPB + j < NGH) issue_piece(ln, (tn - 1) & 1, (q - Q0) * PPB + j, goff, gv);
                }
                if (l >= 1 && q == QM) bar();
                if (L > 1 && l == 1 && q >= QIN && q < QIN + XBLK) {
#pragma unroll
                    for (int j = 0; j < EPB; ++j)
                        if ((q - QIN) * EPB + j < NE) stage_elem((q - QIN) * EPB + j);
                    if (q == QIN + XBLK - 1 && ph + 2 < T) fetch_x(ph + 2);
                }
                if (last && q >= QC && ready) {
#pragma unroll
                    for (int j = 0; j < CPB; ++j)
                        if ((q - QC) * CPB + j < NGH) commit_piece(ln, (q - QC) * CPB + j, gv);
                    if (q == nblk - 1) new_done = true;
                }
            };

            // ---- stacked-gate product on the matrix cores ------------------------------------------------
            f32x4 acc[NMT];
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt) acc[mt] = bias_r[l];
            const float* rec_src = hbuf + (l * MR + r) * SH + 4 * g;
            if (l == 0) {
                layer_mfma<NMT, QX, QX + QH, NW0, ACC_V>(acc, xin + r * SX + 4 * g, SX, rec_src, SH, w0, t > 0, hook);
            } else {
                const float* in_src = (DROP ? dbuf : hbuf) + ((l - 1) * MR + r) * SH + 4 * g;
                if (l == 1) {
                    if constexpr (L > 1) layer_mfma<NMT, QH, 2 * QH, NW1, ACC_V>(acc, in_src, SH, rec_src, SH, w1, t > 0, hook);
                } else {
                    if constexpr (L > 2) layer_mfma<NMT, QH, 2 * QH, NW1, ACC_V>(acc, in_src, SH, rec_src, SH, w2, t > 0, hook);
                }
            }
            mfma_drain();
            raise_pending();                             // (a section too short to reach block QF)
            STAMP_END(2);                                // 2: MFMA layer-step

            // first step of a layer above 0: there was no recurrent span, hence none of the work scheduled in it
            // (the mid-section barrier, in the input span, did run: no reader of xin or hbuf[0] is left)
            bool odd_path = false;
            if (l >= 1 && t == 0) {
                if (l == 1) {
                    stage_x();
                    if (ph + 2 < T) fetch_x(ph + 2);
                }
                odd_path = true;
            }
#ifdef APE_CLUSTER_STAMPS
            if (pre && !ready) st_acc[l == 0 ? 0 : 11] += 1;    // diagnostic: how often the blocking path runs
#endif
            if (peek_pending) { peek_wait(peeked); peek_pending = false; }     // (a short section: looked, never judged)
            if (L > 1 && pre && !ready) {                // first step of a layer, or a peer was late: blocking path
                wait_flags(ln, (unsigned)tn, peeked);
                issue_gather(ln, (tn - 1) & 1, gv);
            }
            STAMP_END(3);                                // 3: flag wait + gather issue of the blocking path

            // ---- gates + cell update, lane-local: registers 0..3 = i,f,g,o of (unit g, batch row 16*mt + r) -------
            {
                const int unit = member * 16 + wave * 4 + g;
#pragma unroll
                for (int mt = 0; mt < NMT; ++mt) {
                    const int row = 16 * mt + r;
                    float hval;
                    if (diag_noact) {
                        const float c = acc[mt][1] * cst[l][mt] + acc[mt][0] * acc[mt][2];
                        cst[l][mt] = c;
                        hval = acc[mt][3] * c;
                    } else {
                        const float iv = gate_act(acc[mt][0], false), fv = gate_act(acc[mt][1], false);
                        const float gv_ = gate_act(acc[mt][2], true), ov = gate_act(acc[mt][3], false);
                        const float c = fv * cst[l][mt] + iv * gv_;
                        cst[l][mt] = c;
                        hval = ov * gate_act(c, true);
                    }
                    own[row * SO + wave * 4 + g] = hval;
                    if (DROP && l < L - 1) {
                        float m;
                        const int b = row0 + row;
                        if (drop_masks) {
                            m = (b < p.B) ? p.masks[(((size_t)l * p.B + b) * T + t) * H + unit] : 0.0f;
                        } else {
                            // same counters as the batch-tile kernel (group of 4 rows, index row & 3): same masks
                            uint32_t rnd[4];
                            // (row_base: the launch's first row in the caller's batch -- a multiple of 16 -- so that a batch served by
                            //  several launches, or partly by the batch-tile kernel, draws the masks of ONE call over its global rows)
                            philox4x32((uint32_t)((p.row_base + b) & ~3), (uint32_t)t, (uint32_t)unit, (uint32_t)l, (uint32_t)p.seed,
                                       (uint32_t)(p.seed >> 32), rnd);
                            const float uf = (float)(rnd[b & 3] >> 8) * (1.0f / 16777216.0f);
                            m = (uf >= p.dropout_p) ? 1.0f / (1.0f - p.dropout_p) : 0.0f;
                        }
                        own[(MR + row) * SO + wave * 4 + g] = hval * m;
                    }
                }
            }
            STAMP_END(4);                                // 4: activations + cell update + own-slice write
            // ---- publish: each wave sends the 16-byte pieces of ITS four units (row = lane; its own LDS staging
            //      columns, no barrier) write-through, notes the flag it owes and moves on
            {
                const int v = lane / MR, row = lane - v * MR;        // variant 1 = masked slice
                if (!diag_noex && lane < ((DROP && l < L - 1) ? 2 * MR : MR)) {
                    const f32x4 hv = *reinterpret_cast<const f32x4*>(own + (v * MR + row) * SO + 4 * wave);
                    const unsigned s_off = (unsigned)((((member * 4 + wave) * MR + row) * 4) * sizeof(float));
                    if (in_l2)
                        __builtin_amdgcn_raw_buffer_store_b128(
                            __builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, hv), hx_rsrc, s_off,
                            hx_base(l, t & 1, v), 0);
                    else
                        __builtin_amdgcn_raw_buffer_store_b128(
                            __builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, hv), hx_rsrc, s_off,
                            hx_base(l, t & 1, v), 16 /* sc1: write-through */);
                    // all-steps mode (DropoutLSTM.forward returns every step, nn_models.py:188-189): the top layer's raw
                    // output of this step also goes to [B,T,H]; the head runs over those rows afterwards (ape_head_rows)
                    if (p.hseq != nullptr && l == L - 1 && v == 0 && row0 + row < p.B)
                        *reinterpret_cast<f32x4*>(p.hseq + ((size_t)(row0 + row) * T + t) * H + member * 16 + 4 * wave) = hv;
                }
                pend_idx = l;
                pend_epoch = (unsigned)(t + 1);
            }
            STAMP_END(6);                                // 6: publish store issue
            if (L == 1) {
                // one layer: publish -> own flag -> the peers' flags -> gather; a barrier (every wave is through its spans: hbuf[0] and
                // xin have no reader left), commit + the next step's x, the end barrier
                if (pre) {
                    raise_pending();
                    wait_flags(0, (unsigned)tn, 0u);
                    issue_gather(0, (tn - 1) & 1, gv);
                }
                bar();
                if (pre) commit_gather(0, gv);
                if (ph + 1 < T) {
                    stage_x();
                    if (ph + 2 < T) fetch_x(ph + 2);
                }
                bar();
                if (ctl[0] != 0) return;
            } else if (last) {
                if (pre && !new_done) commit_gather(ln, gv);      // blocking path / first step: every wave is past
                STAMP_END(7);                                        // the input span (mid or odd-path barrier)
                bar();                                   // end of phase: hbuf[0] and x of the next phase visible
                if (ctl[0] != 0) return;
                STAMP_END(10);                           // 10: end-of-phase barrier
            } else {
                carry = pre;
                if (odd_path) bar();                     // x staged on the odd path must be visible to the next phase
            }
            if (pre) prefetched = true;
            if (L > 1 && l == 0 && ph == 0 && T > 1) {   // x_1 (section 1, which stages x_{ph+1}, idles in phase 0)
                bar();
                stage_x();
                if (T > 2) fetch_x(2);
                bar();
            }
        }
    }
#ifdef APE_CLUSTER_STAMPS
    if (blockIdx.x == 0 && tid == 0) {
        unsigned long long* dbg = reinterpret_cast<unsigned long long*>(p.status + 8);
        for (int k = 0; k < 12; ++k) dbg[k] = st_acc[k];
    }
#endif
    if ((p.flags & APE_DIAG_STAMP) && blockIdx.x == 0 && tid == 0) {
        const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        unsigned long long* dbg = reinterpret_cast<unsigned long long*>(p.status + 4);
        dbg[0] = c1 - stamp_c0;
        dbg[1] = r1 - stamp_r0;
    }
#ifdef APE_CLUSTER_STAMPS
    wg_t2 = __builtin_amdgcn_s_memrealtime();
#endif
    raise_pending();
    // ---- head: each member finishes RPM = MR/GH (>= 1) of the cluster's windows and gathers only THOSE rows of
    //      h^{L-1}_{T-1} (GH x RPM x 4 sixteen-byte pieces = at most one per thread, not the whole 64 KB slice set)
    //      (all-steps callers -- y == nullptr: the head runs over the sequence afterwards, or not at all -- skip it: nobody waits for
    //       the last step's flags then, and the wait below keeps this wave's flag store out of the last workgroup's re-zeroing)
    //      (only the one-layer form, launch A of a bank, takes the short cut: the all-steps launches of whole models keep the wait for the
    //       last step's flags in front of the departure counter, as before round 5 -- ADVICE r05)
    if (L == 1 && p.y == nullptr) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        constexpr int RPM = (MR + GH - 1) / GH;          // rows per member
        static_assert(GH * RPM * 4 <= 256, "one head piece per thread");
        // this thread's share of the head weights first: the loads fly while the last flags and slices arrive
        const int n_out = RPM * O;
        constexpr int PL = (RPM * APE_MAX_OUTPUT * 4 <= 256) ? 4 : ((RPM * APE_MAX_OUTPUT * 2 <= 256) ? 2 : 1);
        constexpr int NHW = H / (4 * PL);                 // 16-byte weight pieces per thread
        const int oi = tid / PL, part = tid % PL;
        const bool live = oi < n_out;
        const int rr = live ? oi / O : 0, o = live ? oi - rr * O : 0;
        const int row = member * RPM + rr;
        f32x4 hw[NHW];
#pragma unroll
        for (int i = 0; i < NHW; ++i)
            hw[i] = (live && row < MR) ? *reinterpret_cast<const f32x4*>(p.w_out + (size_t)o * H + 4 * part + 4 * PL * i)
                                       : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        wait_flags(L - 1, (unsigned)T, 0u);
        const int h_m = tid / (RPM * 4), h_rr = (tid / 4) % RPM, h_quad = tid & 3;
        const int h_row = member * RPM + h_rr;
        if (!diag_noex && tid < GH * RPM * 4 && h_row < MR) {
            const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                hx_rsrc, (unsigned)((((h_m * 4 + h_quad) * MR + h_row) * 4) * sizeof(float)), hx_base(L - 1, (T - 1) & 1, 0),
                16 /* sc1 */));
            *reinterpret_cast<f32x4*>(hbuf + ((L - 1) * MR + h_row) * SH + h_m * 16 + 4 * h_quad) = v;
        }
        bar();
        if (ctl[0] != 0) return;
        // (row, target) dot products over H, PL lanes each (k interleaved by 4), combined by lane shuffles
        float s_acc = 0.0f;
        if (live && row < MR) {
            const float* hv = hbuf + ((L - 1) * MR + row) * SH + 4 * part;
#pragma unroll
            for (int i = 0; i < NHW; ++i) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(hv + 4 * PL * i);
                s_acc = fmaf(a[0], hw[i][0], s_acc); s_acc = fmaf(a[1], hw[i][1], s_acc);
                s_acc = fmaf(a[2], hw[i][2], s_acc); s_acc = fmaf(a[3], hw[i][3], s_acc);
            }
        }
        if (PL >= 2) s_acc += __shfl_xor(s_acc, 1, 64);
        if (PL >= 4) s_acc += __shfl_xor(s_acc, 2, 64);
        const int b = row0 + row;
        if (p.y != nullptr && live && part == 0 && row < MR && b < p.B) p.y[(size_t)b * O + o] = s_acc + p.b_out[o];
    }
#ifdef APE_CLUSTER_STAMPS
    if (tid == 0 && p.dbg_wg != nullptr && blockIdx.x < 256) {
        unsigned long long* d = p.dbg_wg + blockIdx.x * 8;
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        d[0] = (unsigned long long)ticket; d[1] = xcc & 0xF; d[2] = wg_t0; d[3] = wg_t1; d[4] = wg_t2;
        d[5] = __builtin_amdgcn_s_memrealtime();
    }
#endif
    // ---- self-cleaning: the LAST workgroup to get here re-zeroes every polled word for the next launch ------
    // (no memset node in front of the launch: graph replays and back-to-back calls find a clean state).  All
    // the other workgroups are past their last flag read when they count themselves done.  A launch that
    // aborted on an expired spin skips this; ape_model_check() then resets the block from the host.
    __syncthreads();
    if (tid == 0)
        ctl[2] = (__hip_atomic_fetch_add(p.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) ? 1 : 0;
    __syncthreads();
    if (ctl[2] != 0) {
        const int n_words = (int)(gridDim.x / GH) * L * NFL;
        for (int i = tid; i < n_words; i += 256)
            __hip_atomic_store(p.xflags + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cls_mode) {
            for (int i = tid; i < (int)gridDim.x; i += 256) __hip_atomic_store(xcc_words + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (tid < 8) __hip_atomic_store(class_ticket + tid * 16, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (tid == 0) {
            __hip_atomic_store(p.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(p.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

template <int H, int L, int KX, int NMT, bool DROP>
size_t smem_bytes() {
    constexpr int MR = 16 * NMT, NV = DROP ? 2 : 1;
    return ((size_t)L * MR * (H + 8) + (size_t)MR * (KX + 8) + (size_t)NV * MR * 20 +
            (size_t)(DROP ? (L - 1) * MR * (H + 8) : 0) + 4 * 64 + 4 + 4) * sizeof(float);
}

template <int H, int L, int KX, int NMT, bool DROP>
hipError_t launch(const ClusterParams& p, int clusters, hipStream_t stream) {
    const size_t smem = smem_bytes<H, L, KX, NMT, DROP>();
    hipLaunchKernelGGL((ape_lstm_cluster<H, L, KX, NMT, DROP>), dim3(clusters * (H / 16)), dim3(256), smem, stream, p);
    return hipGetLastError();
}

template <int H, int L, int KX, int NMT, bool DROP>
hipError_t prepare() {
    if (smem_bytes<H, L, KX, NMT, DROP>() > APE_LDS_BYTES) return hipErrorInvalidValue;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&ape_lstm_cluster<H, L, KX, NMT, DROP>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, APE_LDS_BYTES);
}

}  // namespace

// shapes the cluster kernel is built for: the deployed pocket / watch-only (H=256, L=2, I<=32) and
// upper-arm (H=128, L=3, 32<I<=64) regressors, and ImuPoseLSTM's 2 x 256 LSTM behind its 256-wide input layer (KX = 256:
// 128 + 128 weight registers per lane = the whole accumulator file, one or two row tiles, no dropout -- its Monte-Carlo
// mode is the plain forward, nn_models.py:246-251)
bool ape_cluster_supported(int H, int L, int KX) {
    return (H == 256 && L == 2 && (KX == 32 || KX == 256)) || (H == 128 && L == 3 && KX == 64);
}
// layer 0 of a deployed shape on its own (launch A of a Monte-Carlo bank): ONE instantiation each, no dropout, two row tiles.  Not a
// model shape: a user-built one-layer LSTM of these widths is served by the batch-tile kernel (the full-model predicate
// must not answer for it, or ape_model_create takes the whole cluster set-up for a model it has no instantiations for)
bool ape_cluster_layer0_supported(int H, int KX) { return (H == 128 && KX == 64) || (H == 256 && KX == 32); }

#define APE_CL_DISPATCH(FN, ...)                                                     \
    if (H == 256 && L == 2 && KX == 32) {                                            \
        if (nmt == 1) return dropout ? FN<256, 2, 32, 1, true>(__VA_ARGS__) : FN<256, 2, 32, 1, false>(__VA_ARGS__); \
        if (nmt == 2) return dropout ? FN<256, 2, 32, 2, true>(__VA_ARGS__) : FN<256, 2, 32, 2, false>(__VA_ARGS__); \
        if (nmt == 4 && !dropout) return FN<256, 2, 32, 4, false>(__VA_ARGS__);      \
    } else if (H == 256 && L == 2 && KX == 256) {                                    \
        if (nmt == 1 && !dropout) return FN<256, 2, 256, 1, false>(__VA_ARGS__);     \
        if (nmt == 2 && !dropout) return FN<256, 2, 256, 2, false>(__VA_ARGS__);     \
    } else if (H == 128 && L == 3 && KX == 64) {                                     \
        if (nmt == 1) return dropout ? FN<128, 3, 64, 1, true>(__VA_ARGS__) : FN<128, 3, 64, 1, false>(__VA_ARGS__); \
        if (nmt == 2) return dropout ? FN<128, 3, 64, 2, true>(__VA_ARGS__) : FN<128, 3, 64, 2, false>(__VA_ARGS__); \
        if (nmt == 4 && !dropout) return FN<128, 3, 64, 4, false>(__VA_ARGS__);      \
    } else if (H == 128 && L == 1 && KX == 64) {  /* layer 0 of the 3 x 128 model on its own: launch A of its Monte-Carlo bank */ \
        if (nmt == 2 && !dropout) return FN<128, 1, 64, 2, false>(__VA_ARGS__);      \
    } else if (H == 256 && L == 1 && KX == 32) {  /* ... and of the 2 x 256 models, for banks of up to 512 streams */ \
        if (nmt == 2 && !dropout) return FN<256, 1, 32, 2, false>(__VA_ARGS__);      \
    }                                                                                \
    return hipErrorInvalidValue;

hipError_t ape_prepare_lstm_cluster(int H, int L, int KX) {
    if (L == 1 ? !ape_cluster_layer0_supported(H, KX) : !ape_cluster_supported(H, L, KX)) return hipErrorInvalidValue;
    for (int nmt : {1, 2, 4})
        for (bool dropout : {false, true}) {
            if (dropout && nmt == 4) continue;
            if (KX == 256 && (dropout || nmt == 4)) continue;       // not built (LDS: 64 rows x 264 columns of x alone)
            if (L == 1 && (dropout || nmt != 2)) continue;          // the layer-0 form: one instantiation
            hipError_t e = [&]() -> hipError_t { APE_CL_DISPATCH(prepare) }();
            if (e != hipSuccess) return e;
        }
    return hipSuccess;
}

hipError_t ape_launch_lstm_cluster(int H, int L, int KX, int nmt, bool dropout, int clusters, const ClusterParams& p,
                                   hipStream_t stream) {
    APE_CL_DISPATCH(launch, p, clusters, stream)
}
