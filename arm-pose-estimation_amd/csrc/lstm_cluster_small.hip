// Small-batch (1..4 windows) variant of the weight-stationary cluster LSTM kernel: the latency path
// (BASELINE.json configs[1]: batch=1 streaming at 50 Hz).
//
// At 1-4 windows a 16-row MFMA tile is 75-94 % padding, so this variant keeps the cluster idea (one workgroup per CU, member m
// owns a slice of the hidden units of every layer, its weights resident in registers for the whole launch, h exchanged
// through memory every phase) but
//   * the stacked-gate product is a register-resident GEMV on the VALU: lane (column c = unit*4+gate, k-group g) multiplies
//     its 4 weights of every k-block with the matching activations (one broadcast ds_read_b128 per block and row); the
//     k-groups are summed with a DPP row rotation and gfx950's row / half swaps (v_permlane16_swap, v_permlane32_swap) --
//     no trip through the LDS crossbar;
//   * lane group g then owns batch row g: one activation per lane, the four gates of a unit meet by DPP quad broadcasts;
//   * ALL layers of a phase are computed together (layer l works on step p - l) -- in the steady state as straight-line
//     code, so the layers' dependent chains overlap -- and there is ONE exchange per phase;
//   * the cluster is H/8 members (UW = 2 units per wave: every CU of a 32-CU XCD at H = 256) where an XCD has that many CUs:
//     100 weight registers per lane, all architectural; H/16 members (UW = 4) otherwise;
//   * the exchange is ONE hop: the lane that holds a fresh h value stores the 8-byte granule {h, tag} (tag = launch number
//     of the model << 12 | phase + 1); every thread polls its own pair of granules (16 bytes, L1-bypassing) until both carry
//     the awaited tag and puts the two values into the LDS buffer of the next phase (h is double-buffered in LDS by phase
//     parity, in memory by step parity).  The data is the flag: no store drain, no flag store, no copy after the poll.  The
//     launch number lives in device memory (bumped by the last member out), so a captured launch replays correctly.
// Same arithmetic as the other kernels up to float32 summation order.
#include <type_traits>
#include "ape_internal.h"
#include "lstm_latency_common.h"
#include "fk_device.h"
#include "../../include/ape_hip.h"

namespace {

// UW = hidden units per wave: 4 (GH = H/16 members; the register layout of the MFMA cluster kernel) or 2 (GH = H/8 members
// = every CU of a 32-CU XCD at H = 256: half the weights, FMAs and LDS reads per lane and phase -- the phase is bound by the
// instruction count at one wave per SIMD -- and a 100-register weight set that stays in the architectural file).
template <int H, int L, int KX, int NR, int UW>
__global__ __launch_bounds__(256, 1) void ape_lstm_cluster_small(const ClusterParams p) {
    constexpr int GH = H / (4 * UW);
    constexpr int CW = 4 * UW, NG = 64 / CW, KB = 4 * NG;   // columns (unit*4 + gate) per wave; k-groups; k-block depth
    constexpr int SH = H, SX = KX + 8;       // LDS row strides: h rows unpadded (broadcast reads: no bank conflicts to pad away)
    constexpr int QX = KX / KB, QH = H / KB;
    constexpr int NW0 = (KX + H) / NG, NW1 = (2 * H) / NG;
    static_assert(NR <= MR && (H == 128 || H == 256), "built for the deployed hidden sizes");
    static_assert((UW == 4 || UW == 2) && KX % KB == 0 && GH >= MR && GH <= 64, "cluster shape");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane % CW, g = lane / CW;        // column = unit*4 + gate; k-group, later batch row
    const int gate = c & 3, u = c >> 2;
    const int T = p.T, I = p.I, O = p.O;
    const bool normalize = (p.flags & APE_FLAG_NORMALIZE_INPUT) != 0;
    const bool bcast_x = (p.flags & APE_FLAG_BROADCAST_X) != 0;   // monte_carlo_predictions: one window, B rows

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* hbuf = smem;                            // [2][L][MR][SH]  by phase parity: a phase reads one, its exchange fills the other
    float* xin = hbuf + 2 * L * MR * SH;           // [2][MR][SX]  double-buffered by step parity
    int* ctl = reinterpret_cast<int*>(xin + 2 * MR * SX);
    // Membership is fixed by the block index: the launch has 8 x GH workgroups and only every eighth one takes part
    // (the others leave at once).  Under the placement observed on this hardware -- blocks are dealt round-robin over
    // the 8 XCDs -- those GH workgroups share ONE XCD, hence one L2, and the exchange can stay inside it: plain stores
    // (acknowledged by the L2, not written through to memory) and L1-bypassing loads, ~0.8 us less per phase.  HIP
    // guarantees no placement, so nothing is assumed: every member publishes the XCD it really runs on (a non-zero
    // word; the last member out zeroes the words again), all members read all GH words once the
    // weights are in, and only if they agree is the in-L2 form used; otherwise the write-through form that is valid for any placement.  One cluster, one launch: a member
    // that is dispatched late delays the launch, it cannot deadlock it.
    if (blockIdx.x % 8 != 0) return;
#ifdef APE_CLUSTER_STAMPS
    unsigned long long st_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long st_begin = st_t0, st_rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    const int member = blockIdx.x / 8;
    const int row0 = 0;
    unsigned my_xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(my_xcc));
    my_xcc &= 0xFu;
    // the bias of this lane's gate row: requested ahead of the weights and consumed (see below) before the phase loop, so that
    // the loop carries no pending load of it -- otherwise the first use in the loop waits for everything but the newest
    // memory operation, i.e. for the x fetch issued at the top of the same phase
    float bias_r[L];
#pragma unroll
    for (int l = 0; l < L; ++l) bias_r[l] = p.bias[l][gate * H + (member * 4 + wave) * UW + u];
    // ---- weights: registers for the whole launch (same fragment layout as the MFMA cluster kernel); requested before
    //      anything else -- everything up to the first phase runs under their latency
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
    // the launch number of this model's latency kernel (bumped by the last member out): the upper bits of every tag
    const unsigned seq = __hip_atomic_load(p.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0xFFFFFu;
    // a launch that finds the sticky status word set (an earlier launch on this model aborted and left stale epochs
    // behind) leaves without touching anything; ape_model_check reports and resets
    if (tid == 0) {
        ctl[0] = (__hip_atomic_load(p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) ? 1 : 0;
        if (ctl[0] == 0)
            __hip_atomic_store(p.xcc_slots + member, 0x10u | my_xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    for (int i = tid; i < 2 * L * MR * SH; i += 256) hbuf[i] = 0.0f;   // h_{-1} = 0: the first step of a layer reads zeros
    // (barriers that wait for LDS traffic only: the weight loads stay in flight)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (ctl[0] != 0) return;

    // head weights of this thread (16 lanes per target, H / 16 consecutive k each): requested in the last phase, under its exchange
    const int hw_o = tid >> 4, hw_part = tid & 15;
    const bool hw_live = member < MR && hw_o < O;
    f32x4 hw[H / 64];
    float hw_b = 0.0f;
#pragma unroll
    for (int i = 0; i < H / 64; ++i) hw[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    float cst[L];
#pragma unroll
    for (int l = 0; l < L; ++l) cst[l] = 0.0f;

    const __amdgpu_buffer_rsrc_t hx_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.hx, 0, (int)p.hx_bytes, 0x00020000);
    const unsigned long long hx_addr = reinterpret_cast<unsigned long long>(p.hx);
    u32x4 hx_desc;                                 // the same descriptor as a scalar tuple for the polling loads' asm
    hx_desc[0] = __builtin_amdgcn_readfirstlane((unsigned)hx_addr);
    hx_desc[1] = __builtin_amdgcn_readfirstlane((unsigned)(hx_addr >> 32) & 0xFFFFu);
    hx_desc[2] = __builtin_amdgcn_readfirstlane((unsigned)p.hx_bytes);
    hx_desc[3] = 0x00020000u;
    // exchange: 8-byte granules {h, tag} at [layer][step parity][row][unit]; tag = launch number << 12 | phase + 1.  A granule
    // is written with one 8-byte store and is valid exactly when its tag is the awaited one, so the data IS the flag: one
    // hop (store -> polled load) per phase instead of store -> acknowledge -> flag -> poll -> copy.
    constexpr unsigned ROW_BYTES = H * 8;                               // a window row of one layer
    constexpr unsigned SET_BYTES = MR * ROW_BYTES;                      // one (layer, parity)
    auto hx_base = [&](int l, int par) -> unsigned { return (unsigned)((l * 2 + par) * SET_BYTES); };
    constexpr int PAIRS = NR * H / 2;                                   // 16-byte pairs of granules per layer
    constexpr int NI = (L * PAIRS + 255) / 256;                         // pairs this thread collects per phase
    int it_l[NI], it_lds[NI];
    unsigned it_off[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int e = tid + 256 * i;
        const int l = e / PAIRS, r = e - l * PAIRS, row = r / (H / 2), pair = r - row * (H / 2);
        it_l[i] = (e < L * PAIRS) ? l : -1;
        it_lds[i] = (l * MR + row) * SH + 2 * pair;
        it_off[i] = hx_base(l, 0) + (unsigned)(row * ROW_BYTES + pair * 16);
    }

    // x_t: f64 z-score, cast f32 (estimator.py:103-104) -- (x - m) / s correctly rounded via the host-rounded reciprocal
    // and one residual step (bit-identical to the division, as in lstm_cluster.hip); fetched a phase ahead
    const int xrow = tid / KX, xk = tid - xrow * KX;
    const bool x_live = tid < MR * KX && xk < I && row0 + xrow < p.B;
    const double x_mean = (normalize && x_live) ? p.xx_m[xk] : 0.0;
    const double x_std = (normalize && x_live) ? p.xx_s[xk] : 1.0;
    const double x_rstd = (normalize && x_live) ? p.xx_r[xk] : 1.0;
    const float* const x_src = p.x + (x_live ? (size_t)(bcast_x ? 0 : row0 + xrow) * T * I + xk : (size_t)0);
    float xr = 0.0f;
    auto fetch_x = [&](int t) {
        if (x_live) xr = x_src[(size_t)(t + p.x_ring >= T ? t + p.x_ring - T : t + p.x_ring) * I];
    };
    auto stage_x = [&](int t) {
        if (tid < MR * KX) {
            const double d = (double)xr - x_mean;
            const double q0 = d * x_rstd;
            const double rr = fma(-q0, x_std, d);
            const double q1 = fma(rr, x_rstd, q0);
            xin[((t & 1) * MR + xrow) * SX + xk] = x_live ? (float)((rr == rr) ? q1 : q0) : 0.0f;
        }
    };
    fetch_x(0);
    stage_x(0);
    if (T > 1) fetch_x(1);
    // ---- do all members really share an XCD?  (their words have been on the way since the weights were requested)
    if (wave == 0) {
        unsigned spins = 0, v = 0u;
        while (true) {
            v = 0x10u | my_xcc;
            if (lane < GH) v = __hip_atomic_load(p.xcc_slots + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__all((int)(v != 0u))) break;
            if (++spins > SPIN_LIMIT || __hip_atomic_load(p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                if (lane == 0) {
                    ctl[0] = 1;
                    __hip_atomic_store(p.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        const int same = __all((int)((v & 0xFu) == my_xcc));
        if (lane == 0) ctl[3] = same;
    }
    // (a barrier that waits for LDS traffic only: the weight loads of the layers above may still be in flight, phase 0 needs
    //  layer 0's only)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (ctl[0] != 0) return;
    // uniform over the cluster: every member read the same GH words
    const bool in_l2 = ctl[3] != 0 && (p.flags & APE_FLAG_ANY_PLACEMENT) == 0;

    const int P = T + L - 1;
#pragma unroll
    for (int l = 0; l < L; ++l) asm volatile("" :: "v"(bias_r[l]));     // the bias has arrived (it was requested first)
    SM_STAMP(0);                                    // 0: prologue (weights into registers, x_0, XCD rendezvous)
#pragma unroll 1
    for (int ph = 0; ph < P; ++ph) {
        // x of the next step: registers (fetched a phase ago) -> the other xin buffer (its readers finished a phase ago), and
        // the fetch of the step after it goes out now, under this phase's compute
        if (ph + 1 < T) stage_x(ph + 1);
        if (ph + 2 < T) fetch_x(ph + 2);
        // ---- every layer of this phase (layer l works on step t = ph - l).  In the steady state (all layers active) the body
        //      is straight-line code: all GEMVs, then all k-group sums, then all cell updates, so that the layers' dependent
        //      chains (LDS read -> FMA chain -> lane sums -> exp/rcp chains) overlap instead of following each other
        const float* const hrd = hbuf + (ph & 1) * (L * MR * SH);
        const unsigned want = (seq << 12) | (unsigned)(ph + 1);
        auto layers = [&](auto all_tag) {
            constexpr bool ALL = decltype(all_tag)::value;
            f32x4 part[L][NR];
#pragma unroll
            for (int l = 0; l < L; ++l)
#pragma unroll
                for (int m = 0; m < NR; ++m) part[l][m] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            if constexpr (ALL && NR == 1 && L == 2) {
                // one row, two layers, steady state: requests of the next span under the multiplications of the last
                // (layer 1's input IS layer 0's recurrent input: h of layer 0 from the last phase, fetched once)
                f32x4 ax[QX], ah[QH], bh[QH];
                span_load<QX, KB>(ax, xin + (ph & 1) * MR * SX + 4 * g);
                span_load<QH, KB>(ah, hrd + 4 * g);
                span_load<QH, KB>(bh, hrd + MR * SH + 4 * g);
                __builtin_amdgcn_sched_barrier(0);
                span_fma<QX, NW0>(part[0][0], ax, w0, 0);
                span_fma<QH, NW0>(part[0][0], ah, w0, 4 * QX);
                span_fma<QH, NW1>(part[1][0], ah, w1, 0);
                span_fma<QH, NW1>(part[1][0], bh, w1, 4 * QH);
            } else
#pragma unroll
            for (int l = 0; l < L; ++l) {
                const int t = ph - l;
                if (!ALL && (t < 0 || t >= T)) continue;         // uniform
                const float* rec_src = hrd + l * MR * SH + 4 * g;
                if (l == 0) {
                    gemv_span<NR, QX, KB, NW0>(part[l], xin + (t & 1) * MR * SX + 4 * g, SX, w0, 0);
                    gemv_span<NR, QH, KB, NW0>(part[l], rec_src, SH, w0, 4 * QX);
                } else {
                    const float* in_src = hrd + (l - 1) * MR * SH + 4 * g;
                    if (l == 1) {
                        if constexpr (L > 1) {
                            gemv_span<NR, QH, KB, NW1>(part[l], in_src, SH, w1, 0);
                            gemv_span<NR, QH, KB, NW1>(part[l], rec_src, SH, w1, 4 * QH);
                        }
                    } else {
                        if constexpr (L > 2) {
                            gemv_span<NR, QH, KB, NW1>(part[l], in_src, SH, w2, 0);
                            gemv_span<NR, QH, KB, NW1>(part[l], rec_src, SH, w2, 4 * QH);
                        }
                    }
                }
            }
            // sum the NG k-groups (lanes c, c + CW, ...): row rotation / row swap / half swap, no LDS crossbar
            float pre[L];
#pragma unroll
            for (int l = 0; l < L; ++l) {
                pre[l] = 0.0f;
#pragma unroll
                for (int m = 0; m < NR; ++m) {
                    float v = (part[l][m][0] + part[l][m][1]) + (part[l][m][2] + part[l][m][3]);
                    if constexpr (CW == 8) v = sum_ror8(v);
                    v = sum_xor16(v);
                    v = sum_xor32(v);
                    if (g == m) pre[l] = v;            // lane group g owns batch row g from here on
                }
            }
#pragma unroll
            for (int l = 0; l < L; ++l) {
                const int t = ph - l;
                if (!ALL && (t < 0 || t >= T)) continue;
                const float a = gate_act(pre[l] + bias_r[l], gate == 2);
                const float iv = quad_bcast<0>(a), fv = quad_bcast<1>(a), gv = quad_bcast<2>(a), ov = quad_bcast<3>(a);
                const float cn = fv * cst[l] + iv * gv;
                cst[l] = cn;
                // publish: the lane that holds h of (row g, unit) sends its granule {h, tag} -- fire and forget
                const float hval = ov * gate_act(cn, true);
                u32x2 gran;
                gran[0] = __builtin_bit_cast(unsigned, hval);
                gran[1] = want;
                const unsigned off = (gate == 0 && g < NR)
                                         ? hx_base(l, t & 1) + (unsigned)(g * ROW_BYTES + ((member * 4 + wave) * UW + u) * 8) : 0x80000000u;
                if (in_l2) __builtin_amdgcn_raw_buffer_store_b64(gran, hx_rsrc, off, 0, 0);      // stays in the XCD's L2
                else __builtin_amdgcn_raw_buffer_store_b64(gran, hx_rsrc, off, 0, 16 /* sc1: write-through */);
            }
        };
        if (ph >= L - 1 && ph < T) layers(std::true_type{});
        else layers(std::false_type{});
        if (ph == P - 1 && hw_live) {              // the head's weights: their latency hides behind this phase's exchange
#pragma unroll
            for (int i = 0; i < H / 64; ++i)
                hw[i] = *reinterpret_cast<const f32x4*>(p.w_out + (size_t)hw_o * H + hw_part * (H / 16) + 4 * i);
            hw_b = p.b_out[hw_o];
        }
        SM_STAMP(1);                                // 1: x staging + GEMVs + gates + granule stores of every layer
        // ---- collect: every thread polls ITS pairs of granules (16 bytes: two units of one row and layer) until both carry this
        //      phase's tag, then puts the two values into the h buffer of the next phase
        {
            unsigned off[NI];
            bool act[NI];
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int t = ph - it_l[i];
                act[i] = it_l[i] >= 0 && t >= 0 && t < T;
                off[i] = act[i] ? it_off[i] + (unsigned)(t & 1) * SET_BYTES : 0x80000000u;   // out of range: reads zeros
            }
            // (the four words of a pair leave the vector inside the loop, one by one: building the float pair for the LDS
            //  write from elements 0 and 2 of the asm's vector result after the loop is miscompiled by this hipcc -- it
            //  writes element 0 twice)
            unsigned val0[NI], val1[NI];
            unsigned spins = 0;
            while (true) {
                u32x4 v[NI];
                poll_granules<NI>(v, off, hx_desc);
                bool bad = false;
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    val0[i] = v[i][0];
                    val1[i] = v[i][2];
                    bad = bad || (act[i] && (v[i][1] != want || v[i][3] != want));
                }
                if (!__any((int)bad)) break;
                // (the sticky status word is looked at every 256th spin only: a second dependent load per spin doubles the
                //  time a late granule costs)
                if (++spins > SPIN_LIMIT || ((spins & 255u) == 0u &&
                                             __hip_atomic_load(p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                    if (lane == 0) {
                        ctl[0] = 1;
                        __hip_atomic_store(p.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    break;
                }
                if (spins > 64u) __builtin_amdgcn_s_sleep(1);
            }
            SM_STAMP(3);                            // 3: last granule store issued -> every awaited granule seen
            float* const hwr = hbuf + ((ph + 1) & 1) * (L * MR * SH);
#pragma unroll
            for (int i = 0; i < NI; ++i)
                if (act[i]) {
                    hwr[it_lds[i]] = __builtin_bit_cast(float, val0[i]);
                    hwr[it_lds[i] + 1] = __builtin_bit_cast(float, val1[i]);
                }
        }
        SM_STAMP(4);                                // 4: values into LDS
        __syncthreads();                            // gathered h and x_{ph+1} visible
        SM_STAMP(5);                                // 5: barrier
        if (ctl[0] != 0) return;
    }

    // ---- head: member m finishes row m (rows < NR <= 4 <= GH): 16 lanes per target (16 k each), combined by lane shuffles;
    //      this thread's share of the head weights was requested at the top of the kernel
    if (member < MR) {
        const int b = row0 + member;
        float s_acc = 0.0f;
        if (hw_live) {
            const float* hv = hbuf + (P & 1) * (L * MR * SH) + ((L - 1) * MR + member) * SH + hw_part * (H / 16);
#pragma unroll
            for (int i = 0; i < H / 64; ++i) {
                const f32x4 av = *reinterpret_cast<const f32x4*>(hv + 4 * i);
                s_acc = fmaf(av[0], hw[i][0], s_acc); s_acc = fmaf(av[1], hw[i][1], s_acc);
                s_acc = fmaf(av[2], hw[i][2], s_acc); s_acc = fmaf(av[3], hw[i][3], s_acc);
            }
        }
        // the 16 lanes of a target: quad swaps, half-row mirror, row mirror (DPP)
        s_acc += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s_acc), 0xB1, 0xF, 0xF, false));
        s_acc += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s_acc), 0x4E, 0xF, 0xF, false));
        s_acc += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s_acc), 0x141, 0xF, 0xF, false));
        s_acc += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s_acc), 0x140, 0xF, 0xF, false));
        if (hw_live && hw_part == 0 && b < p.B) p.y[(size_t)b * O + hw_o] = s_acc + hw_b;
        // ---- the post-filter of this row in the same launch (ape_infer): de-normalise in f64 (estimator.py:108-109), then the row's
        //      three chains side by side as fk.hip's ape_fk3_kernel has them -- wave 0, lanes 0 / 1 the lower / upper arm's 6D ->
        //      quaternion and rotated bone, wave 1 the hips quaternion and shoulder origin, the two joining sums behind a barrier.
        //      The same arithmetic on the same float32 targets: bit-identical to the two-launch form, one launch gap shorter.
        if (p.fk_est != nullptr) {                  // (uniform over the launch; `member < MR` is uniform over the workgroup)
            using namespace ape_fkdev;
            __shared__ double fk_y[20];
            __shared__ double fk_rot[3][3];
            if (hw_live && hw_part == 0) {
                double v = (double)(s_acc + hw_b);
                if (p.fk_yy_m) {
#pragma clang fp contract(off)
                    v = v * p.fk_yy_s[hw_o] + p.fk_yy_m[hw_o];
                }
                fk_y[hw_o] = v;
            }
            __syncthreads();
            const bool hips = p.fk_layout != APE_LAYOUT_ORI_CAL_LARM_UARM;
            const bool full = p.fk_layout == APE_LAYOUT_ORI_POS_CAL_LARM_UARM_HIPS;
            const int c_l = full ? 3 : 0, c_u = full ? 12 : 6, c_h = full ? 18 : 12;
            const int e_lq = hips ? 9 : 6, e_uq = hips ? 13 : 10, e_hq = 17;
            auto put = [&](int c, double v) {
                if (p.fk_est_dtype == APE_F32) static_cast<float*>(p.fk_est)[(size_t)b * p.fk_W + c] = (float)v;
                else static_cast<double*>(p.fk_est)[(size_t)b * p.fk_W + c] = v;
            };
            const int fw = tid >> 6, fl = tid & 63;
            if (b < p.B) {
                if (fw == 0 && fl < 2) {
                    const Quat q = six_drr_to_quat(fk_y + (fl ? c_u : c_l));
                    const Vec3 bone = fl ? Vec3{p.fk_body[3], p.fk_body[4], p.fk_body[5]} : Vec3{p.fk_body[0], p.fk_body[1], p.fk_body[2]};
                    const Vec3 v = qrot(q, bone);
                    fk_rot[fl][0] = v.x; fk_rot[fl][1] = v.y; fk_rot[fl][2] = v.z;
                    const int eq = fl ? e_uq : e_lq;
                    put(eq, q.w); put(eq + 1, q.x); put(eq + 2, q.y); put(eq + 3, q.z);
                } else if (fw == 1 && fl == 0) {
                    Vec3 uo{p.fk_body[6], p.fk_body[7], p.fk_body[8]};
                    if (hips) {
                        const Quat hq = hips_quat(fk_y[c_h], fk_y[c_h + 1]);
                        uo = qrot(hq, uo);
                        put(e_hq, hq.w); put(e_hq + 1, hq.x); put(e_hq + 2, hq.y); put(e_hq + 3, hq.z);
                        put(6, uo.x); put(7, uo.y); put(8, uo.z);
                    }
                    fk_rot[2][0] = uo.x; fk_rot[2][1] = uo.y; fk_rot[2][2] = uo.z;
                } else if (fw == 2 && fl < 6 && full) {      // hand and lower-arm positions are network outputs (estimate_joints.py:20-45)
                    put(fl, fk_y[fl < 3 ? fl : 6 + fl]);
                }
            }
            __syncthreads();
            if (b < p.B && !full && fw == 0 && fl < 6) {
#pragma clang fp contract(off)
                const int k = fl % 3;
                const double lo = fk_rot[1][k] + fk_rot[2][k];                  // qrot(uq, uarm_vec) + uo
                if (fl >= 3) put(3 + k, lo);
                else put(k, fk_rot[0][k] + lo);                                 // qrot(lq, larm_vec) + lo
            }
        }
    }
    SM_STAMP(6);                                    // 6: head
#ifdef APE_CLUSTER_STAMPS
    if (p.dbg_wg != nullptr && tid == 0 && member == 0) {
        for (int k = 0; k < 8; ++k) p.dbg_wg[k] = st_acc[k];
        p.dbg_wg[8] = __builtin_amdgcn_s_memtime() - st_begin;
        p.dbg_wg[9] = __builtin_amdgcn_s_memrealtime() - st_rt0;
    }
#endif
    __syncthreads();
    if (tid == 0)
        ctl[2] = (__hip_atomic_fetch_add(p.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == GH - 1) ? 1 : 0;
    __syncthreads();
    if (ctl[2] != 0) {
        // the next launch's tags differ from every tag of this one; when the 20-bit launch number wraps, the granules go
        // back to zero (tag 0 is never awaited) so that a tag of 2^20 launches ago cannot be taken for a fresh one
        if (seq == 0xFFFFFu)
            for (int i = tid; i < (int)(p.hx_bytes / 4); i += 256)
                __hip_atomic_store(reinterpret_cast<unsigned*>(p.hx) + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid == 0) __hip_atomic_store(p.seq, seq + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid < GH) __hip_atomic_store(p.xcc_slots + tid, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid == 0) {
            __hip_atomic_store(p.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(p.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

template <int H, int L, int KX, int NR, int UW>
hipError_t launch_small(const ClusterParams& p, hipStream_t stream) {
    constexpr size_t smem = ((size_t)2 * L * MR * H + 2 * MR * (KX + 8) + 4) * sizeof(float);
    hipLaunchKernelGGL((ape_lstm_cluster_small<H, L, KX, NR, UW>), dim3(8 * (H / (4 * UW))), dim3(256), smem, stream, p);
    return hipGetLastError();
}

template <int H, int L, int KX, int UW>
hipError_t launch_small_nr(int nr, const ClusterParams& p, hipStream_t stream) {
    if (nr == 1) return launch_small<H, L, KX, 1, UW>(p, stream);
    if (nr == 2) return launch_small<H, L, KX, 2, UW>(p, stream);
    if (nr == 4) return launch_small<H, L, KX, 4, UW>(p, stream);
    return hipErrorInvalidValue;
}

}  // namespace

// one cluster, B <= 4 windows (nr = 1, 2 or 4 rows computed); uw = hidden units per wave (4: H/16 members, weights p.wcl in
// the MFMA cluster kernel's order; 2: H/8 members, weights in the [member][wave][k-quad][lane][4] order of 8-column waves)
hipError_t ape_launch_lstm_cluster_small(int H, int L, int KX, int nr, int uw, const ClusterParams& p, hipStream_t stream) {
    if (H == 256 && L == 2 && KX == 32) return uw == 2 ? launch_small_nr<256, 2, 32, 2>(nr, p, stream) : launch_small_nr<256, 2, 32, 4>(nr, p, stream);
    if (H == 128 && L == 3 && KX == 64) return uw == 2 ? launch_small_nr<128, 3, 64, 2>(nr, p, stream) : launch_small_nr<128, 3, 64, 4>(nr, p, stream);
    return hipErrorInvalidValue;
}
