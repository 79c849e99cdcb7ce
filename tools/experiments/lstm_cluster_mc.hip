// Latency kernel for ONE stream in Monte-Carlo mode: the estimators' default frame (watch_phone_pocket_nn.py:13-19,
// nn_models.py:191-207: n dropout samples of one window, x.repeat((n,1,1)) through a train-mode 2-layer LSTM), n <= 64.
//
// Built like lstm_cluster_small.hip -- 32 members = every CU of one XCD, member m owns hidden units [8m, 8m+8) of both
// layers with its weights in registers for the whole launch, 8-byte tagged granules {value, tag} through the XCD's L2 as the
// only hand-over -- with the two things this mode allows:
//   * nn.LSTM's dropout sits BETWEEN the layers, so layer 0 sees the same input and the same state for every sample: it is
//     computed ONCE per step (the VALU GEMV of the small-batch kernel, one row).  The member that produced h_0(t)[unit]
//     also applies the n masks (Philox with the counters of every other kernel of this library: same seed, same masks; or
//     the caller's injected masks) and publishes the n masked values beside the plain one;
//   * layer 1 is a real GEMM (n rows): v_mfma_f32_16x16x4_f32 with the member's 32 gate rows as two A tiles ordered
//     unit*4 + gate (128 weight registers per lane in the accumulator file) and the samples as 16-row B tiles read from LDS;
//     wave w takes A tile w & 1 and the B tiles (w >> 1), (w >> 1) + 2, ...  Every lane ends up with the four gates of one
//     (unit, sample) cell in its four accumulator registers: the cell update needs no lane exchange.
// Phase ph: layer 0 works on step ph, layer 1 on step ph - 1; one collect per phase (every thread polls 16-byte pairs of
// granules in rounds of 16 loads and writes the values into LDS), two workgroup barriers (the activation tile is single-
// buffered: at 64 samples it is 133 KB).  Same arithmetic as the other kernels up to float32 summation order.
#include <type_traits>
#include "ape_internal.h"
#include "lstm_latency_common.h"
#include "../../include/ape_hip.h"

namespace {

constexpr int MC_H = 256, MC_KX = 32, MC_GH = 32;
constexpr int MC_SA = 2 * MC_H + 8;           // row stride of the layer-1 activation tile: [h_0 masked | h_1] + pad
constexpr int MC_RND = 16;                    // polled pairs per thread and round

// v_mfma_f32_16x16x4_f32 with A = a weight that lives in an AGPR for the whole launch (row lane & 15 = unit * 4 + gate of the
// wave's A tile), B = an activation (column lane & 15 = sample) and the accumulator in the accumulator file; hipcc does not
// model an asm MFMA's hazards: start values settle under `mc_settle`, results are read after `mc_drain` (lstm_cluster_common.h)
__device__ __forceinline__ void mc_mfma(f32x4& acc, float act, float w) {
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %2, %1, %0" : "+a"(acc) : "v"(act), "a"(w));
}
__device__ __forceinline__ void mc_settle() { asm volatile("s_nop 4" ::: "memory"); }
__device__ __forceinline__ void mc_drain() { asm volatile("s_nop 15\n\ts_nop 7" ::: "memory"); }

// 16 polling loads (16 bytes per lane, L1-bypassing), 4 KiB apart (pair e = i * 256 + tid), and the wait for them in ONE
// statement; the scalar offset walks in M0
__device__ __forceinline__ void poll_round16(u32x4 (&v)[MC_RND], unsigned voff, u32x4 rsrc, unsigned sbase) {
#define MC_LD(i) "buffer_load_dwordx4 %" #i ", %[vo], %[rs], m0 offen sc1\n\ts_add_u32 m0, m0, 0x1000\n\t"
    asm volatile("s_mov_b32 m0, %[sb]\n\t" MC_LD(0) MC_LD(1) MC_LD(2) MC_LD(3) MC_LD(4) MC_LD(5) MC_LD(6) MC_LD(7) MC_LD(8) MC_LD(9)
                 MC_LD(10) MC_LD(11) MC_LD(12) MC_LD(13) MC_LD(14) MC_LD(15) "s_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]),
                   "=&v"(v[8]), "=&v"(v[9]), "=&v"(v[10]), "=&v"(v[11]), "=&v"(v[12]), "=&v"(v[13]), "=&v"(v[14]), "=&v"(v[15])
                 : [vo] "v"(voff), [rs] "s"(rsrc), [sb] "s"(sbase)
                 : "memory", "scc");
#undef MC_LD
}

template <int NT>
__global__ __launch_bounds__(256, 1) void ape_lstm_cluster_mc(const ClusterParams p) {
    constexpr int H = MC_H, KX = MC_KX, GH = MC_GH, SA = MC_SA;
    constexpr int NROW = 16 * NT, NT2 = (NT + 1) / 2;      // sample rows (padded); B tiles per wave
    constexpr int SX = KX + 8, KB = 32, QX = KX / KB, QH = H / KB, NW0 = (KX + H) / 8;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = p.T, I = p.I, O = p.O, n = p.B;
    const bool normalize = (p.flags & APE_FLAG_NORMALIZE_INPUT) != 0;
    const bool inj_masks = (p.flags & APE_FLAG_DROPOUT_MASKS) != 0;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* a1 = smem;                              // [NROW][SA]  layer 1's B operand: h_0(t) masked | h_1(t-1)
    float* h0buf = a1 + NROW * SA;                 // [2][H]  layer 0's own state, by phase parity
    float* xin = h0buf + 2 * H;                    // [2][SX]
    int* ctl = reinterpret_cast<int*>(xin + 2 * SX);
    if (blockIdx.x % 8 != 0) return;               // membership by block index, placement verified below (lstm_cluster_small.hip)
    const int member = blockIdx.x / 8;
#ifdef APE_CLUSTER_STAMPS
    unsigned long long st_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long st_begin = st_t0, st_rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    unsigned my_xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(my_xcc));
    my_xcc &= 0xFu;

    // ---- layer 0 (GEMV): lane = (k-group g, column c = unit * 4 + gate) of this wave's two units ---------------------------
    const int c = lane & 7, g = lane >> 3, gate = c & 3, u = c >> 2;
    const int unit0 = (member * 4 + wave) * 2 + u;
    const float bias0 = p.bias[0][gate * H + unit0];
    // ---- layer 1 (MFMA): A tile mt = wave & 1 (units 8 m + 4 mt + 0..3), lane = (sample column nn, unit ug)
    const int mt = wave & 1, nt0 = wave >> 1, nn = lane & 15, ug = lane >> 4;
    const int unit1 = member * 8 + mt * 4 + ug;
    f32x4 bias1;
#pragma unroll
    for (int i = 0; i < 4; ++i) bias1[i] = p.bias[1][i * H + unit1];
    float w0[NW0];                                 // 36: [k-quad][lane][4] of the H/8-member latency layout
    {
        const f32x4* s0 = reinterpret_cast<const f32x4*>(p.wcl[0]) + ((size_t)(member * 4 + wave) * (NW0 / 4)) * 64 + lane;
#pragma unroll
        for (int i = 0; i < NW0 / 4; ++i) {
            const f32x4 v = s0[i * 64];
            w0[4 * i] = v[0]; w0[4 * i + 1] = v[1]; w0[4 * i + 2] = v[2]; w0[4 * i + 3] = v[3];
        }
    }
    float wa[128];                                 // layer 1: A fragments, register 4 q + j = Wcat1[row][16 q + 4 (lane >> 4) + j]
    {
        const f32x4* s1 = reinterpret_cast<const f32x4*>(p.wcl[1]) + ((size_t)(member * 2 + mt) * 32) * 64 + lane;
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const f32x4 v = s1[i * 64];
            wa[4 * i] = v[0]; wa[4 * i + 1] = v[1]; wa[4 * i + 2] = v[2]; wa[4 * i + 3] = v[3];
        }
    }
    const unsigned seq = __hip_atomic_load(p.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0xFFFFFu;
    if (tid == 0) {
        ctl[0] = (__hip_atomic_load(p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) ? 1 : 0;
        if (ctl[0] == 0)
            __hip_atomic_store(p.xcc_slots + member, 0x10u | my_xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    for (int i = tid; i < (NROW * SA + 2 * H) / 4; i += 256) reinterpret_cast<f32x4*>(smem)[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (ctl[0] != 0) return;

    const __amdgpu_buffer_rsrc_t hx_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.hx, 0, (int)p.hx_bytes, 0x00020000);
    const unsigned long long hx_addr = reinterpret_cast<unsigned long long>(p.hx);
    u32x4 hx_desc;
    hx_desc[0] = __builtin_amdgcn_readfirstlane((unsigned)hx_addr);
    hx_desc[1] = __builtin_amdgcn_readfirstlane((unsigned)(hx_addr >> 32) & 0xFFFFu);
    hx_desc[2] = __builtin_amdgcn_readfirstlane((unsigned)p.hx_bytes);
    hx_desc[3] = 0x00020000u;
    // granules of one phase parity, in pair order: [h_0: 128 pairs][h_0 masked: n x 128][h_1: n x 128]; pair = 16 bytes
    const unsigned pairs_m = 128u, pairs_1 = 128u + (unsigned)n * 128u, pairs_all = 128u + 2u * (unsigned)n * 128u;
    const unsigned par_bytes = (128u + 2u * 64u * 128u) * 16u;
    const int n_rounds = (int)((pairs_all + 256u * MC_RND - 1u) / (256u * MC_RND));

    // x_t of the one window: f64 z-score, cast f32 (estimator.py:103-104), a phase ahead (as lstm_cluster_small.hip)
    const bool x_live = tid < KX && tid < I;
    const double x_mean = (normalize && x_live) ? p.xx_m[tid] : 0.0;
    const double x_std = (normalize && x_live) ? p.xx_s[tid] : 1.0;
    const double x_rstd = (normalize && x_live) ? p.xx_r[tid] : 1.0;
    float xr = 0.0f;
    auto fetch_x = [&](int t) {
        if (x_live) xr = p.x[(size_t)(t + p.x_ring >= T ? t + p.x_ring - T : t + p.x_ring) * I + tid];
    };
    auto stage_x = [&](int t) {
        if (tid < KX) {
            const double d = (double)xr - x_mean;
            const double q0 = d * x_rstd;
            const double rr = fma(-q0, x_std, d);
            const double q1 = fma(rr, x_rstd, q0);
            xin[(t & 1) * SX + tid] = x_live ? (float)((rr == rr) ? q1 : q0) : 0.0f;
        }
    };
    fetch_x(0);
    stage_x(0);
    if (T > 1) fetch_x(1);
    if (wave == 0) {                               // do all members share an XCD?
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
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (ctl[0] != 0) return;
    const bool in_l2 = ctl[3] != 0 && (p.flags & APE_DIAG_WRITE_THROUGH) == 0;
    auto store_granule = [&](float val, unsigned tag, unsigned off) {
        u32x2 gran;
        gran[0] = __builtin_bit_cast(unsigned, val);
        gran[1] = tag;
        if (in_l2) __builtin_amdgcn_raw_buffer_store_b64(gran, hx_rsrc, off, 0, 0);
        else __builtin_amdgcn_raw_buffer_store_b64(gran, hx_rsrc, off, 0, 16 /* sc1: write-through */);
    };

    float c0 = 0.0f, c1[NT2];
#pragma unroll
    for (int i = 0; i < NT2; ++i) c1[i] = 0.0f;
    asm volatile("" :: "v"(bias0), "v"(bias1));
    const float keep_scale = 1.0f / (1.0f - p.dropout_p);

    SM_STAMP(0);                                    // 0: prologue
#pragma unroll 1
    for (int ph = 0; ph <= T; ++ph) {
        const bool l0 = ph < T, l1 = ph >= 1;
        const unsigned want = (seq << 12) | (unsigned)(ph + 1);
        const unsigned pbase = (unsigned)(ph & 1) * par_bytes;
        if (ph + 1 < T) stage_x(ph + 1);
        if (ph + 2 < T) fetch_x(ph + 2);
        // ---- layer 0, step ph: one row, every lane of a unit's quad group ends up with the unit's h ----------------------
        if (l0) {
            const float* hrd = h0buf + (ph & 1) * H;
            f32x4 ax[QX], ah[QH];
            span_load<QX, KB>(ax, xin + (ph & 1) * SX + 4 * g);
            span_load<QH, KB>(ah, hrd + 4 * g);
            f32x4 part = {0.0f, 0.0f, 0.0f, 0.0f};
            span_fma<QX, NW0>(part, ax, w0, 0);
            span_fma<QH, NW0>(part, ah, w0, 4 * QX);
            float v = (part[0] + part[1]) + (part[2] + part[3]);
            v = sum_ror8(v);
            v = sum_xor16(v);
            v = sum_xor32(v);
            const float a = gate_act(v + bias0, gate == 2);
            const float iv = quad_bcast<0>(a), fv = quad_bcast<1>(a), gv = quad_bcast<2>(a), ov = quad_bcast<3>(a);
            const float cn = fv * c0 + iv * gv;
            c0 = cn;
            const float h0 = ov * gate_act(cn, true);
            if (gate == 0 && g == 0) store_granule(h0, want, pbase + (unsigned)unit0 * 8u);
            // the n masked copies of this unit's value: lane = (unit u, sample b = gate bits | k-group bits), b + 32 for n > 32
#pragma unroll
            for (int half = 0; half < (NT > 2 ? 2 : 1); ++half) {
                const int b = (gate | (g << 2)) + 32 * half;
                float m;
                if (inj_masks) {
                    m = (b < n) ? p.masks[((size_t)b * T + ph) * H + unit0] : 0.0f;
                } else {
                    uint32_t rnd[4];
                    philox4x32((uint32_t)(b & ~3), (uint32_t)ph, (uint32_t)unit0, 0u, (uint32_t)p.seed, (uint32_t)(p.seed >> 32), rnd);
                    const float uf = (float)(rnd[b & 3] >> 8) * (1.0f / 16777216.0f);
                    m = (uf >= p.dropout_p) ? keep_scale : 0.0f;
                }
                store_granule(h0 * m, want, (b < n) ? pbase + (pairs_m * 16u) + (unsigned)(b * H + unit0) * 8u : 0x80000000u);
            }
        }
        SM_STAMP(1);                                // 1: x staging + layer 0 (GEMV, gates, masks, granule stores)
        // ---- layer 1, step ph - 1: 16 gate rows x 16 samples x K = 512 per tile ------------------------------------------
        if (l1 && (NT > 1 || nt0 == 0)) {
#pragma unroll
            for (int ti = 0; ti < NT2; ++ti) {
                const int nt = nt0 + 2 * ti;
                f32x4 acc = bias1;
                const float* src = a1 + (16 * nt + nn) * SA + 4 * ug;
                f32x4 bcur = *reinterpret_cast<const f32x4*>(src), bnxt = bcur;
                mc_settle();
#pragma unroll
                for (int q = 0; q < 32; ++q) {
                    if (q + 1 < 32) bnxt = *reinterpret_cast<const f32x4*>(src + 16 * (q + 1));
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) mc_mfma(acc, bcur[j], wa[4 * q + j]);
                    __builtin_amdgcn_sched_barrier(0);
                    bcur = bnxt;
                }
                mc_drain();
                const float iv = gate_act(acc[0], false), fv = gate_act(acc[1], false);
                const float gv = gate_act(acc[2], true), ov = gate_act(acc[3], false);
                const float cn = fv * c1[ti] + iv * gv;
                c1[ti] = cn;
                const float h1 = ov * gate_act(cn, true);
                const int b = 16 * nt + nn;
                store_granule(h1, want, (b < n) ? pbase + pairs_1 * 16u + (unsigned)(b * H + unit1) * 8u : 0x80000000u);
            }
        }
        SM_STAMP(2);                                // 2: layer 1 (MFMAs, cell update, granule stores)
        __syncthreads();                            // every wave is done reading the activation tile
        if (ctl[0] != 0) return;
        SM_STAMP(3);                                // 3: first barrier
        // ---- collect: pair e = i * 256 + tid of this phase's parity; awaited = the arrays whose layer was active ------------
        {
            float* const h0wr = h0buf + ((ph + 1) & 1) * H;
            for (int r = 0; r < n_rounds; ++r) {
                unsigned val0[MC_RND], val1[MC_RND];
                bool act[MC_RND];
                unsigned spins = 0;
#pragma unroll
                for (int i = 0; i < MC_RND; ++i) {
                    const unsigned e = (unsigned)((r * MC_RND + i) * 256 + tid);
                    act[i] = e < pairs_all && ((e < pairs_1) ? l0 : l1);
                }
                while (true) {
                    u32x4 v[MC_RND];
                    poll_round16(v, (unsigned)tid * 16u, hx_desc, pbase + (unsigned)(r * MC_RND) * 4096u);
                    bool bad = false;
#pragma unroll
                    for (int i = 0; i < MC_RND; ++i) {
                        val0[i] = v[i][0];
                        val1[i] = v[i][2];
                        bad = bad || (act[i] && (v[i][1] != want || v[i][3] != want));
                    }
                    if (!__any((int)bad)) break;
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
#pragma unroll
                for (int i = 0; i < MC_RND; ++i) {
                    if (!act[i]) continue;
                    const unsigned e = (unsigned)((r * MC_RND + i) * 256 + tid);
                    float* dst;
                    if (e < pairs_m) dst = h0wr + 2 * e;
                    else if (e < pairs_1) dst = a1 + ((e - pairs_m) >> 7) * SA + 2 * ((e - pairs_m) & 127u);
                    else dst = a1 + ((e - pairs_1) >> 7) * SA + H + 2 * ((e - pairs_1) & 127u);
                    dst[0] = __builtin_bit_cast(float, val0[i]);
                    dst[1] = __builtin_bit_cast(float, val1[i]);
                }
            }
        }
        SM_STAMP(4);                                // 4: collect (poll rounds + LDS writes)
        __syncthreads();                            // the next phase's operands are in LDS
        if (ctl[0] != 0) return;
        SM_STAMP(5);                                // 5: second barrier
    }

    // ---- head: member m finishes samples m and m + 32 from h_1(T-1) in the activation tile; 16 lanes per target --------------
    {
        const int hw_o = tid >> 4, hw_part = tid & 15;
        for (int b = member; b < n; b += GH) {
            float s_acc = 0.0f;
            if (hw_o < O) {
                const float* hv = a1 + b * SA + H + hw_part * (H / 16);
                const float* wv = p.w_out + (size_t)hw_o * H + hw_part * (H / 16);
#pragma unroll
                for (int i = 0; i < H / 64; ++i) {
                    const f32x4 av = *reinterpret_cast<const f32x4*>(hv + 4 * i);
                    const f32x4 wq = *reinterpret_cast<const f32x4*>(wv + 4 * i);
                    s_acc = fmaf(av[0], wq[0], s_acc); s_acc = fmaf(av[1], wq[1], s_acc);
                    s_acc = fmaf(av[2], wq[2], s_acc); s_acc = fmaf(av[3], wq[3], s_acc);
                }
            }
            s_acc += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s_acc), 0xB1, 0xF, 0xF, false));
            s_acc += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s_acc), 0x4E, 0xF, 0xF, false));
            s_acc += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s_acc), 0x141, 0xF, 0xF, false));
            s_acc += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s_acc), 0x140, 0xF, 0xF, false));
            if (hw_o < O && hw_part == 0) p.y[(size_t)b * O + hw_o] = s_acc + p.b_out[hw_o];
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
    if (ctl[2] != 0) {                              // last member out (as lstm_cluster_small.hip)
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

template <int NT>
constexpr size_t mc_smem() { return ((size_t)16 * NT * MC_SA + 2 * MC_H + 2 * (MC_KX + 8) + 8) * sizeof(float); }

template <int NT>
hipError_t launch_mc(const ClusterParams& p, hipStream_t stream) {
    hipLaunchKernelGGL((ape_lstm_cluster_mc<NT>), dim3(8 * MC_GH), dim3(256), mc_smem<NT>(), stream, p);
    return hipGetLastError();
}

}  // namespace

bool ape_cluster_mc_supported(int H, int L, int KX) { return H == MC_H && L == 2 && KX == MC_KX; }

// bytes of the granule buffer (two phase parities, up to 64 samples)
size_t ape_cluster_mc_granule_bytes() { return (size_t)2 * (128 + 2 * 64 * 128) * 16; }

hipError_t ape_prepare_lstm_cluster_mc() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ape_lstm_cluster_mc<1>), hipFuncAttributeMaxDynamicSharedMemorySize, APE_LDS_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ape_lstm_cluster_mc<2>), hipFuncAttributeMaxDynamicSharedMemorySize, APE_LDS_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ape_lstm_cluster_mc<4>), hipFuncAttributeMaxDynamicSharedMemorySize, APE_LDS_BYTES);
    return e;
}

// one stream, n = p.B <= 64 dropout samples of ONE window (p.x is that window), last-step output [n, O]
hipError_t ape_launch_lstm_cluster_mc(const ClusterParams& p, hipStream_t stream) {
    if (p.B <= 16) return launch_mc<1>(p, stream);
    if (p.B <= 32) return launch_mc<2>(p, stream);
    if (p.B <= 64) return launch_mc<4>(p, stream);
    return hipErrorInvalidValue;
}
