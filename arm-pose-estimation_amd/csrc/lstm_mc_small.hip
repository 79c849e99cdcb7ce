// Latency kernel of the Monte-Carlo frame: ONE stream (or up to eight), n dropout samples of its window -- the frame every deployed
// estimator runs (watch_phone_pocket_nn.py:13-19 -> nn_models.py:191-207: lstm.train(), x.repeat((n,1,1)), last step), n <= 128.
//
// Built on lstm_cluster_small.hip's frame (weights resident in registers for the launch, a member = a CU owns 8 hidden units of
// every layer, 8-byte tagged granules {value, tag} through ONE XCD's L2 as the only hand-over) with what this mode allows:
//   * the SAMPLES ARE INDEPENDENT, so they are dealt over the EIGHT XCDs: the launch is 8 clusters (cluster = blockIdx % 8 = one
//     XCD under the round-robin placement, verified at run time like the other kernels), each runs the whole network for its
//     R <= 16 sample rows with its own copy of the weights in its CUs' registers.  Nothing crosses an XCD; the exchange of a
//     cluster carries its own rows only (a one-XCD form moved n x 256 granules per layer past every member: 66 us at n = 25);
//   * nn.LSTM's dropout sits BETWEEN the layers (nn_models.py:169-174), so layer 0 sees the same input and state for every
//     sample of a stream: ONE row, the register-resident VALU GEMV of the latency kernel, computed per cluster;
//   * the layers above are GEMMs over the cluster's rows: v_mfma_f32_16x16x4_f32, a member's 32 gate columns as two 16-column
//     A tiles ordered unit * 4 + gate, K = [masked h of the layer below | own h] split over wave pairs (wave = column tile x K
//     half, the halves joined through LDS), the rows as ONE 16-row B tile in LDS -- a lane ends up with the four gates of one
//     (unit, row) cell, the cell update is lane-local;
//   * the dropout masks never travel: a layer's output is published ONCE, plain, and every consumer applies the mask of its
//     rows when it puts the value into its B tile.  The mask multipliers (Philox4x32-10 with the counters of every other kernel
//     of this library -- row quad, step, unit, layer: same seed, same masks -- or the caller's injected masks) do not depend on
//     data, so each phase computes those of the NEXT phase between its publish and its poll, where it would otherwise wait.
// Phase ph: layer l works on step ph - l; one collect and two workgroup barriers per phase.  Same arithmetic as the other kernels
// up to float32 summation order.
//
// The frame's feature builder in the same launch (ape_streams_frame_host, banks of up to eight streams): one extra workgroup per stream
// builds the new feature row from the raw message (parse_device.h, the float64 chain of ape_parse_rows_kernel: ~5 us on one lane)
// BESIDE the clusters' weight prologue, hands it to them as tagged granules (the newest step is the last one the window needs) and
// writes it into the stream's ring for the frames to come: two launches per frame instead of three, the feature builder off the
// frame's critical path (one stream x 25 samples, host in / host out: 49.0 -> 43.3 us).  The post-filter as this launch's tail (last
// workgroup out, targets written through) was built and measured too -- 48.2 us alone, 43.1 with the builder: it buys nothing over
// its own launch (one workgroup's cold float64 chain behind the slowest cluster costs what the launch boundary does) and is not kept.
#include <type_traits>
#include "ape_internal.h"
#include "lstm_latency_common.h"
#include "parse_device.h"
#include "../../include/ape_hip.h"

namespace {

// NI polling loads in one statement (poll_granules covers up to four)
template <int NI>
__device__ __forceinline__ void poll_pairs(u32x4 (&v)[NI], const unsigned (&off)[NI], u32x4 rsrc) {
    if constexpr (NI <= 4) {
        poll_granules<NI>(v, off, rsrc);
    } else {
        static_assert(NI == 5 || NI == 9, "pairs per thread of the built row capacities");
#define MCS_LD(i) "buffer_load_dwordx4 %" #i ", %[o" #i "], %[rs], 0 offen sc1\n\t"
        if constexpr (NI == 5)
            asm volatile("s_nop 4\n\t" MCS_LD(0) MCS_LD(1) MCS_LD(2) MCS_LD(3) MCS_LD(4) "s_waitcnt vmcnt(0)"
                         : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4])
                         : [o0] "v"(off[0]), [o1] "v"(off[1]), [o2] "v"(off[2]), [o3] "v"(off[3]), [o4] "v"(off[4]), [rs] "s"(rsrc)
                         : "memory");
        else
            asm volatile("s_nop 4\n\t" MCS_LD(0) MCS_LD(1) MCS_LD(2) MCS_LD(3) MCS_LD(4) MCS_LD(5) MCS_LD(6) MCS_LD(7) MCS_LD(8) "s_waitcnt vmcnt(0)"
                         : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]),
                           "=&v"(v[8])
                         : [o0] "v"(off[0]), [o1] "v"(off[1]), [o2] "v"(off[2]), [o3] "v"(off[3]), [o4] "v"(off[4]), [o5] "v"(off[5]),
                           [o6] "v"(off[6]), [o7] "v"(off[7]), [o8] "v"(off[8]), [rs] "s"(rsrc)
                         : "memory");
#undef MCS_LD
    }
}

// H, L, KX: the deployed shapes (2 x 256 with KX = 32; 3 x 128 with KX = 64).  RC: row capacity of a cluster (4, 8, 16) -- sizes the
// B tiles, the exchange, the polls and the number of 4-row groups of the matrix product.  INJ: the caller's masks (float
// multipliers, fetched a phase ahead) instead of the in-kernel Philox draws (keep / drop bits for all steps, laid down in the
// weight prologue's shadow).
//
// The layers above layer 0 run on v_mfma_f32_4x4x1_16b_f32: sixteen 4 x 4 outer products per instruction, block = (unit of the member,
// k slice), the four A rows of a block = the unit's four gates, the four B columns = four sample rows -- so a 4-row cluster pays for 4
// rows, not for a 16-row tile (one window x 25 samples: 0.9 -> 0.25 us per phase), and a lane ends up with the four gates of one
// (unit, row) cell.  K = [masked h of the layer below | own h] is split over the four waves (waves 0, 1: the input half; 2, 3: the
// recurrent half, skipped at step 0), two k per instruction (the wave's halves), the partial sums meet in LDS.
template <int H, int L, int KX, int RC, bool INJ, bool FED>
__global__ __launch_bounds__(256, 1) void ape_lstm_mc_small(const McSmallParams p) {
    constexpr int GH = H / 8;                     // members of a cluster
    constexpr int LM = L - 1;                     // layers above layer 0 (MFMA)
    constexpr int NRG = RC / 4;                   // 4-row groups
    constexpr int SA = 2 * H + 8;                 // row stride of an activation tile: [masked input | own h] + pad
    constexpr int SX = KX + 8, KB = 32, QX = KX / KB, QH = H / KB, NW0 = (KX + H) / 8;
    constexpr int NI4 = H / 4;                    // 4x4x1 instructions per wave, layer and row group: a K quarter (H / 2 values), two k each
    constexpr int PH = H / 2;                     // 16-byte granule pairs per row
    constexpr int PAIRS1 = LM * RC * PH;          // pairs of the layers above, one parity: [layer 1: RC rows][layer 2: RC rows]
    constexpr int PAIRS = PH + PAIRS1;            // ... behind the PH pairs of h_0
    constexpr int NG = 256 / PH;                  // every h_0 pair is polled by NG threads: thread (pair, group) fans it out to rows r = group (mod NG)
    constexpr int RG = (RC + NG - 1) / NG;
    constexpr int NI1 = (PAIRS1 + 255) / 256;
    constexpr int NI = 1 + NI1;                   // polled pairs per thread
    constexpr unsigned PAR_BYTES = (unsigned)PAIRS * 16u;
    static_assert(L >= 2 && L <= 3 && (H == 128 || H == 256) && GH <= 32 && (RC == 4 || RC == 8 || RC == 16) && RC <= GH && 256 % PH == 0 &&
                  LM * H == 256, "shape");

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cluster = blockIdx.x & 7, member = blockIdx.x >> 3;
    const int T = p.T, I = p.I, O = p.O;
    const bool normalize = (p.flags & APE_FLAG_NORMALIZE_INPUT) != 0;
    // this cluster's rows: stream = cluster / cps, its `part`-th run of R sample rows
    const int stream = cluster / p.cps, part = cluster - stream * p.cps;
    int rv = (stream < p.n_streams) ? p.n_mc - part * p.R : 0;
    rv = rv < 0 ? 0 : (rv > p.R ? p.R : rv);
    const int row0 = stream * p.n_mc + part * p.R;          // global index of the cluster's first sample row

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* a = smem;                               // [2 phase parity][LM][RC][SA]  B tiles of the MFMA layers
    float* h0buf = a + 2 * LM * RC * SA;           // [2][H]  layer 0's own state
    float* xin = h0buf + 2 * H;                    // [2][SX]
    float* red = xin + 2 * SX;                     // [LM][NRG][4 waves][32][4]  partial sums of the K quarters
    int* ctl = reinterpret_cast<int*>(red + LM * NRG * 4 * 32 * 4);
    float* mv = reinterpret_cast<float*>(ctl + 8); // INJ: [2][LM][RC][H] mask multipliers by phase parity
    unsigned short* mbits = reinterpret_cast<unsigned short*>(ctl + 8);   // else: [T steps][LM * H] (T <= 64: the launcher refuses longer windows) bit r = row r of the cluster keeps (layer, unit) at that step
#ifdef APE_CLUSTER_STAMPS
    unsigned long long st_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long st_begin = st_t0, st_rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    const unsigned seq = __hip_atomic_load(p.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0xFFFFFu;
    const unsigned want_x = (seq << 12) | 0xFFFu;                       // tag of the feature granules (phase tags stay below)
    const bool builder = FED && blockIdx.x >= 8 * GH;                   // host frames: the feature builder of stream blockIdx - 8 GH
    if (builder) {
        __shared__ float prow[64];
        __shared__ double pxx[40];
        const int bs = (int)blockIdx.x - 8 * GH;
        if (tid < p.raw_width) {
            float v = p.raw_rows[(size_t)bs * p.raw_width + tid];
            if (p.raw_big_endian) v = __builtin_bit_cast(float, __builtin_bswap32(__builtin_bit_cast(unsigned, v)));
            prow[tid] = v;
        }
        __syncthreads();
        if (tid == 0) ape_parsedev::parse_row(prow, p.raw_width, p.raw_kind, pxx);
        __syncthreads();
        // for this launch's clusters: {feature, tag} written through (they run on other XCDs)
        if (tid < I) {
            u32x2 gran;
            gran[0] = __builtin_bit_cast(unsigned, (float)pxx[tid]);
            gran[1] = want_x;
            const __amdgpu_buffer_rsrc_t xg_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.xg, 0, 8 * 64 * 8, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b64(gran, xg_rsrc, (unsigned)(bs * 64 + tid) * 8u, 0, 16 /* sc1: write-through */);
        }
        // for the frames to come: the stream's ring (every copy of its window; every slot on a cold start)
        for (int idx = tid; idx < p.ring_rep * I; idx += 256) {
            const int j = idx / I, i = idx - j * I;
            p.ring_out[(size_t)bs * p.ring_stream_stride + (size_t)j * p.ring_rep_stride + i] = (float)pxx[i];
        }
    }
    if (!builder && rv > 0) {
    unsigned my_xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(my_xcc));
    my_xcc &= 0xFu;
    unsigned* const xcc_words = p.xcc_slots + 192 + cluster * 32;

    // ---- layer 0 (GEMV): lane = (k-group g, column c = unit * 4 + gate) of this wave's two units
    const int c = lane & 7, g = lane >> 3, gate = c & 3, u = c >> 2;
    const int unit0 = (member * 4 + wave) * 2 + u;
    const float bias0 = p.bias[0][gate * H + unit0];
    // ---- layers above (4x4x1 MFMA): lane = (block b = lane >> 2: unit ub = b & 7 of the member, k slice ks = b >> 3; q = lane & 3: the
    //      gate on the A side, the row of the group on the B side)
    const int q4 = lane & 3, ub = (lane >> 2) & 7, ks = lane >> 5;
    const int unit1 = member * 8 + ub;
    float w0[NW0];
    {
        const f32x4* s0 = reinterpret_cast<const f32x4*>(p.w0) + ((size_t)(member * 4 + wave) * (NW0 / 4)) * 64 + lane;
#pragma unroll
        for (int i = 0; i < NW0 / 4; ++i) {
            const f32x4 v = s0[i * 64];
            w0[4 * i] = v[0]; w0[4 * i + 1] = v[1]; w0[4 * i + 2] = v[2]; w0[4 * i + 3] = v[3];
        }
    }
    // 16-row clusters (more than 64 samples of one window) take the 16 x 16 x 4 form instead: its 32 MAC per cycle beat the 4x4x1's 16 once
    // the tile is full.  Wave = (column tile ct16 = wave & 1: units 8 m + 4 ct16 + 0..3, K half kh16 = wave >> 1), lane = (row nn16, unit
    // ug16 of the tile); the halves meet in LDS, the waves with kh16 = 0 finish the cells.
    constexpr bool WIDE = RC == 16;
    const int ct16 = wave & 1, kh16 = wave >> 1, nn16 = lane & 15, ug16 = lane >> 4;
    const int unit16 = member * 8 + ct16 * 4 + ug16;
    float wa[LM][NI4];                             // register i = Wcat_l[gate q4 * H + unit1][wave * H/2 + 8 (i / 4) + 4 ks + (i % 4)]
                                                   // (WIDE: register 4 q + j = Wcat_l[(lane & 3) * H + unit16][kh16 * H + 16 q + 4 ug16 + j])
#pragma unroll
    for (int l = 0; l < LM; ++l) {
        const f32x4* s1 = reinterpret_cast<const f32x4*>(WIDE ? p.w16[l + 1] : p.w[l + 1]) + ((size_t)(member * 4 + wave) * (NI4 / 4)) * 64 + lane;
#pragma unroll
        for (int i = 0; i < NI4 / 4; ++i) {
            const f32x4 v = s1[i * 64];
            wa[l][4 * i] = v[0]; wa[l][4 * i + 1] = v[1]; wa[l][4 * i + 2] = v[2]; wa[l][4 * i + 3] = v[3];
        }
    }
    // the cell (layer l, row group rgp) is finished by wave (l * NRG + rgp) & 3, lanes 0..31: lane = (unit ub, row q4 of the group)
    constexpr int NCS = WIDE ? LM : (LM * NRG + 3) / 4;        // cells per lane of a finishing wave
    f32x4 biasc[NCS];
    float cst[NCS];
#pragma unroll
    for (int k = 0; k < NCS; ++k) {
        const int cid = wave + 4 * k, l = WIDE ? k : cid / NRG;
        const bool mine = WIDE ? kh16 == 0 : cid < LM * NRG;
        cst[k] = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) biasc[k][i] = mine ? p.bias[l + 1][i * H + (WIDE ? unit16 : unit1)] : 0.0f;
    }
    // (a store only: nothing here may wait for a load yet -- vmcnt counts in order, the first awaited result waits for every weight)
    if (tid == 0) __hip_atomic_store(xcc_words + member, 0x10u | my_xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // zero state (h_{-1} = 0 of every layer)
    for (int i = tid; i < (2 * LM * RC * SA + 2 * H + 2 * SX) / 4; i += 256)
        reinterpret_cast<f32x4*>(smem)[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    char* const gx = p.gx + (size_t)cluster * p.gx_cluster_bytes;
    const __amdgpu_buffer_rsrc_t hx_rsrc = __builtin_amdgcn_make_buffer_rsrc(gx, 0, (int)p.gx_cluster_bytes, 0x00020000);
    const unsigned long long hx_addr = reinterpret_cast<unsigned long long>(gx);
    u32x4 hx_desc;
    hx_desc[0] = __builtin_amdgcn_readfirstlane((unsigned)hx_addr);
    hx_desc[1] = __builtin_amdgcn_readfirstlane((unsigned)(hx_addr >> 32) & 0xFFFFu);
    hx_desc[2] = __builtin_amdgcn_readfirstlane(p.gx_cluster_bytes);
    hx_desc[3] = 0x00020000u;

    // the pairs this thread collects: slot 0 = h_0 pair pr0 (fanned out to the rows of group rg); slots 1.. = pair e1 = tid + 256 (i - 1)
    // of the layers above -> (layer, row, pair of units)
    const int pr0 = tid % PH, rg = tid / PH;
    int it_l[NI1], it_r[NI1], it_pr[NI1];
#pragma unroll
    for (int i = 0; i < NI1; ++i) {
        const int e1 = tid + 256 * i;
        it_l[i] = (e1 < PAIRS1) ? 1 + e1 / (RC * PH) : -1;
        it_r[i] = (e1 / PH) % RC;
        it_pr[i] = e1 % PH;
    }

    // x_t of the stream's window: f64 z-score, cast f32 (estimator.py:103-104), fetched a phase ahead (as lstm_cluster_small.hip)
    const bool x_live = tid < KX && tid < I;
    const double x_mean = (normalize && x_live) ? p.xx_m[tid] : 0.0;
    const double x_std = (normalize && x_live) ? p.xx_s[tid] : 1.0;
    const double x_rstd = (normalize && x_live) ? p.xx_r[tid] : 1.0;
    const float* const x_src = p.x + (size_t)stream * p.x_stream_stride;
    float xr = 0.0f;
    // host frames: the newest step (every step on a cold start) comes from the builder workgroup's granules, the others from the ring --
    // ONE load instruction either way (descriptor and offset chosen by a uniform select: a load inside a branch makes the compiler
    // wait for everything in flight at the join, which cost 0.2 us per phase when this was an if / else), requested two phases ahead
    // of its use, looked at in stage_x (the builder is long done then) and polled only if it is not
    constexpr bool fed = FED;                   // (its own instantiation: the plain launch pays nothing for the hand-over below)
    const __amdgpu_buffer_rsrc_t xg_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.xg, 0, 8 * 64 * 8, 0x00020000);
    const __amdgpu_buffer_rsrc_t xw_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x_src), 0, T * I * 4, 0x00020000);
    const unsigned xg_off = x_live ? (unsigned)(stream * 64 + tid) * 8u : 0x80000000u;
    u32x2 xg2 = {0u, 0u};
    auto from_builder = [&](int t) { return fed && (p.cold != 0 || t == T - 1); };
    auto fetch_x = [&](int t) {
        const int slot = t + p.x_ring >= T ? t + p.x_ring - T : t + p.x_ring;
        if constexpr (!FED) {
            if (x_live) xr = x_src[(size_t)slot * I + tid];
            return;
        }
        const bool fb = from_builder(t);
        const unsigned off = fb ? xg_off : (x_live ? (unsigned)(slot * I + tid) * 4u : 0x80000000u);
        // (an ordinary load: a granule line this XCD's L2 still holds from the last frame carries the last frame's tag, and the poll
        //  in stage_x -- L2-bypassing -- fetches the fresh one)
        if (x_live) xg2 = __builtin_amdgcn_raw_buffer_load_b64(fb ? xg_rsrc : xw_rsrc, off, 0, 0);
    };
    auto stage_x = [&](int t) {
        if (FED && from_builder(t) && tid < 64) {   // (wave 0: the threads that stage)
            unsigned spins = 0;
            while (__any((int)(x_live && xg2[1] != want_x))) {
                xg2 = __builtin_amdgcn_raw_buffer_load_b64(xg_rsrc, xg_off, 0, 16 /* sc1 */);
                if (++spins > SPIN_LIMIT || ((spins & 255u) == 0u &&
                                             __hip_atomic_load(p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                    if (lane == 0) {
                        ctl[0] = 1;
                        __hip_atomic_store(p.status, 6u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (6: the builder's row)
                    }
                    break;
                }
                if (spins > 16u) __builtin_amdgcn_s_sleep(2);
            }
        }
        if constexpr (FED) xr = __builtin_bit_cast(float, xg2[0]);
        if (tid < KX) {
            const double d = (double)xr - x_mean;
            const double q0 = d * x_rstd;
            const double rr = fma(-q0, x_std, d);
            const double q1 = fma(rr, x_rstd, q0);
            xin[(t & 1) * SX + tid] = x_live ? (float)((rr == rr) ? q1 : q0) : 0.0f;
        }
    };
    // ---- dropout masks.  Thread = (layer l, unit) of the layers that are masked (LM * H = 256 of them).
    //      Philox: the keep bits of layer l's output of step t for the cluster's rows, one call per four rows with the counters of
    //      every other kernel (row quad, step, unit, layer); the first steps here, where the weight loads are in flight and the
    //      VALU has nothing to do, the rest a few steps ahead of their use inside the phases.
    //      Injected: the caller's multipliers of the values collected in phase `target`, fetched a phase ahead.
    const float keep_scale = 1.0f / (1.0f - p.dropout_p);
    const int m_l = tid / H, m_unit = tid - m_l * H;
    const int quads = ((row0 + rv + 3) >> 2) - (row0 >> 2);          // Philox calls per (layer, step, unit)
    auto mask_bits = [&](int t) {
        unsigned word = 0u;
        for (int qb = row0 & ~3; qb < row0 + rv; qb += 4) {
            uint32_t rnd[4];
            philox4x32((uint32_t)qb, (uint32_t)t, (uint32_t)m_unit, (uint32_t)m_l, (uint32_t)p.seed, (uint32_t)(p.seed >> 32), rnd);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = qb + i - row0;
                const float uf = (float)(rnd[i] >> 8) * (1.0f / 16777216.0f);
                if (r >= 0 && r < rv && uf >= p.dropout_p) word |= 1u << r;
            }
        }
        mbits[t * (LM * H) + tid] = (unsigned short)word;
    };
    auto mask_fill = [&](int target) {             // INJ
        const int t = target - m_l;
        if (t < 0 || t >= T) return;
        float* dst = mv + (((target & 1) * LM + m_l) * RC) * H + m_unit;
        for (int r = 0; r < rv; ++r) dst[r * H] = p.masks[(((size_t)m_l * p.rows + row0 + r) * T + t) * H + m_unit];
    };
    // steps laid down ahead: about eight Philox calls per thread in the prologue
    int pre_t = 8 / quads;
    pre_t = pre_t < 1 ? 1 : (pre_t > T ? T : pre_t);
    if constexpr (!INJ) for (int t = 0; t < pre_t; ++t) mask_bits(t);
    // a launch that finds the sticky status word set (an earlier launch on this model aborted) leaves without touching anything
    if (tid == 0) ctl[0] = (__hip_atomic_load(p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) ? 1 : 0;
    if constexpr (INJ) mask_fill(0);
    fetch_x(0);
    stage_x(0);
    if (T > 1) fetch_x(1);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (ctl[0] != 0) return;

    if (wave == 0) {                               // do all members of this cluster share an XCD?
        unsigned spins = 0, v = 0u;
        while (true) {
            v = 0x10u | my_xcc;
            if (lane < GH) v = __hip_atomic_load(xcc_words + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__all((int)(v != 0u))) break;
            if (++spins > SPIN_LIMIT || __hip_atomic_load(p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                if (lane == 0) {
                    ctl[0] = 1;
                    __hip_atomic_store(p.status, 5u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (5: the XCD rendezvous)
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
    const bool in_l2 = ctl[3] != 0 && (p.flags & APE_FLAG_ANY_PLACEMENT) == 0;
    auto store_granule = [&](float val, unsigned tag, unsigned off) {
        u32x2 gran;
        gran[0] = __builtin_bit_cast(unsigned, val);
        gran[1] = tag;
        if (in_l2) __builtin_amdgcn_raw_buffer_store_b64(gran, hx_rsrc, off, 0, 0);      // stays in the XCD's L2
        else __builtin_amdgcn_raw_buffer_store_b64(gran, hx_rsrc, off, 0, 16 /* sc1: write-through */);
    };

    float c0 = 0.0f;
    asm volatile("" :: "v"(bias0));
    const int P = T + L - 1;
    SM_STAMP(0);                                    // 0: prologue
#pragma unroll 1
    for (int ph = 0; ph < P; ++ph) {
        const unsigned want = (seq << 12) | (unsigned)(ph + 1);
        if (ph + 1 < T) stage_x(ph + 1);
        if (ph + 2 < T) fetch_x(ph + 2);
        // ---- layer 0, step ph: one row
        if (ph < T) {
            const float* hrd = h0buf + (ph & 1) * H;
            f32x4 ax[QX], ah[QH];
            span_load<QX, KB>(ax, xin + (ph & 1) * SX + 4 * g);
            span_load<QH, KB>(ah, hrd + 4 * g);
            f32x4 part4 = {0.0f, 0.0f, 0.0f, 0.0f};
            span_fma<QX, NW0>(part4, ax, w0, 0);
            span_fma<QH, NW0>(part4, ah, w0, 4 * QX);
            float v = (part4[0] + part4[1]) + (part4[2] + part4[3]);
            v = sum_ror8(v);
            v = sum_xor16(v);
            v = sum_xor32(v);
            const float av = gate_act(v + bias0, gate == 2);
            const float iv = quad_bcast<0>(av), fv = quad_bcast<1>(av), gv = quad_bcast<2>(av), ov = quad_bcast<3>(av);
            const float cn = fv * c0 + iv * gv;
            c0 = cn;
            const float h0 = ov * gate_act(cn, true);
            store_granule(h0, want, (gate == 0 && g == 0) ? (unsigned)(ph & 1) * PAR_BYTES + (unsigned)unit0 * 8u : 0x80000000u);
        }
        SM_STAMP(1);                                // 1: x staging + layer 0
        // ---- layers above, step ph - l
        f32x4 acc16[LM];
#pragma unroll
        for (int l = 0; l < LM; ++l) {
            const int t = ph - 1 - l;
            if constexpr (WIDE) {                   // this wave's K half of its column tile, 16 rows
                acc16[l] = biasc[l];
                if (t < 0 || t >= T || (kh16 == 1 && t == 0)) continue;    // uniform; h_{-1} = 0: no recurrent half at step 0
                const float* src = a + (((ph & 1) * LM + l) * RC + nn16) * SA + kh16 * H + 4 * ug16;
#pragma unroll
                for (int q = 0; q < H / 16; ++q) {
                    const f32x4 b = *reinterpret_cast<const f32x4*>(src + 16 * q);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc16[l] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[l][4 * q + j], b[j], acc16[l], 0, 0, 0);
                }
                if (kh16 == 1) *reinterpret_cast<f32x4*>(red + ((l * 2 + ct16) * 64 + lane) * 4) = acc16[l];
            } else {                                // this wave's K quarter, every row group
                if (t < 0 || t >= T || (wave >= 2 && t == 0)) continue;    // uniform; h_{-1} = 0: no recurrent half at step 0
                const float* src = a + (((ph & 1) * LM + l) * RC + q4) * SA + wave * (H / 2) + 4 * ks;
                f32x4 acc[NRG][2];
#pragma unroll
                for (int r = 0; r < NRG; ++r) { acc[r][0] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; acc[r][1] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; }
#pragma unroll
                for (int i4 = 0; i4 < NI4 / 4; ++i4) {
                    f32x4 b[NRG];
#pragma unroll
                    for (int r = 0; r < NRG; ++r) b[r] = *reinterpret_cast<const f32x4*>(src + 4 * r * SA + 8 * i4);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int r = 0; r < NRG; ++r)      // (two accumulators per group: a chain of dependent 4x4x1 MFMAs needs wait states)
                            acc[r][j & 1] = __builtin_amdgcn_mfma_f32_4x4x1f32(wa[l][4 * i4 + j], b[r][j], acc[r][j & 1], 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < NRG; ++r) {
                    f32x4 sv;
#pragma unroll
                    for (int i = 0; i < 4; ++i) sv[i] = sum_xor32(acc[r][0][i] + acc[r][1][i]);      // the two k slices of the wave
                    if (lane < 32) *reinterpret_cast<f32x4*>(red + (((l * NRG + r) * 4 + wave) * 32 + lane) * 4) = sv;
                }
            }
        }
        SM_STAMP(2);                                // 2: MFMA spans
        __syncthreads();                            // the partial sums are in LDS
        SM_STAMP(3);
        auto finish_cell = [&](f32x4 sv, int k, int t, int l, int row, int unit) {
            const float iv = gate_act(sv[0], false), fv = gate_act(sv[1], false);
            const float gv = gate_act(sv[2], true), ov = gate_act(sv[3], false);
            const float cn = fv * cst[k] + iv * gv;
            cst[k] = cn;
            const float hv = ov * gate_act(cn, true);
            store_granule(hv, want, (row < rv) ? (unsigned)(t & 1) * PAR_BYTES + (unsigned)(PH + (l * RC + row) * PH) * 16u + (unsigned)unit * 8u
                                               : 0x80000000u);
        };
        if constexpr (WIDE) {
            if (kh16 == 0) {
#pragma unroll
                for (int l = 0; l < LM; ++l) {
                    const int t = ph - 1 - l;
                    if (t < 0 || t >= T) continue;
                    f32x4 sv = acc16[l];
                    if (t > 0) {
                        const f32x4 o = *reinterpret_cast<const f32x4*>(red + ((l * 2 + ct16) * 64 + lane) * 4);
                        sv[0] += o[0]; sv[1] += o[1]; sv[2] += o[2]; sv[3] += o[3];
                    }
                    finish_cell(sv, l, t, l, nn16, unit16);
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < NCS; ++k) {
                const int cid = wave + 4 * k;
                if (cid >= LM * NRG) continue;
                const int l = cid / NRG, r = cid - l * NRG, t = ph - 1 - l;
                if (t < 0 || t >= T) continue;
                if (lane < 32) {
                    f32x4 sv = biasc[k];
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        if (w >= 2 && t == 0) continue;
                        const f32x4 o = *reinterpret_cast<const f32x4*>(red + (((l * NRG + r) * 4 + w) * 32 + lane) * 4);
                        sv[0] += o[0]; sv[1] += o[1]; sv[2] += o[2]; sv[3] += o[3];
                    }
                    finish_cell(sv, k, t, l, 4 * r + q4, unit1);
                }
            }
        }
        SM_STAMP(4);                                // 4: cell updates + publish
        // ---- masks of later steps: data-independent, computed where this phase would otherwise wait for its peers
        if constexpr (INJ) { if (ph + 1 < P) mask_fill(ph + 1); }
        else { if (ph + pre_t < T) mask_bits(ph + pre_t); }
        SM_STAMP(5);                                // 5: masks
        // ---- collect: every thread polls ITS pairs of granules until they carry this phase's tag, then puts the values where the
        //      next phase reads them: own-layer recurrent input as it is, the next layer's input under the rows' masks
        {
            unsigned off[NI];
            bool act[NI];
            act[0] = ph < T;
            off[0] = act[0] ? (unsigned)(ph & 1) * PAR_BYTES + (unsigned)pr0 * 16u : 0x80000000u;
#pragma unroll
            for (int i = 0; i < NI1; ++i) {
                const int t = ph - it_l[i];
                act[1 + i] = it_l[i] >= 1 && t >= 0 && t < T && it_r[i] < rv;
                off[1 + i] = act[1 + i] ? (unsigned)(t & 1) * PAR_BYTES + (unsigned)(PH + tid + 256 * i) * 16u : 0x80000000u;
            }
            // (the multipliers do not depend on the awaited values: fetched in front of the poll, not behind it)
            float* const anext = a + ((ph + 1) & 1) * (LM * RC * SA);
            auto mask2 = [&](int l, int r, int pr) -> f32x2 {      // layer l's output of step ph - l, row r, units 2 pr, 2 pr + 1
                if constexpr (INJ) {
                    return *reinterpret_cast<const f32x2*>(mv + (((ph & 1) * LM + l) * RC + r) * H + 2 * pr);
                } else {
                    const unsigned w2 = *reinterpret_cast<const unsigned*>(mbits + (ph - l) * (LM * H) + l * H + 2 * pr);
                    return f32x2{((w2 >> r) & 1u) ? keep_scale : 0.0f, ((w2 >> (16 + r)) & 1u) ? keep_scale : 0.0f};
                }
            };
            f32x2 m0[RG], m1[NI1];
#pragma unroll
            for (int k = 0; k < RG; ++k) m0[k] = act[0] ? mask2(0, (rg + NG * k) % RC, pr0) : f32x2{0.0f, 0.0f};
#pragma unroll
            for (int i = 0; i < NI1; ++i) m1[i] = (L > 2 && act[1 + i] && it_l[i] < L - 1) ? mask2(it_l[i], it_r[i], it_pr[i]) : f32x2{0.0f, 0.0f};
            unsigned val0[NI], val1[NI];
            unsigned spins = 0;
            while (true) {
                u32x4 v[NI];
                poll_pairs<NI>(v, off, hx_desc);
                bool bad = false;
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    val0[i] = v[i][0];
                    val1[i] = v[i][2];
                    bad = bad || (act[i] && (v[i][1] != want || v[i][3] != want));
                }
                if (!__any((int)bad)) break;
                if (++spins > SPIN_LIMIT || ((spins & 255u) == 0u &&
                                             __hip_atomic_load(p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                    if (lane == 0) {
                        ctl[0] = 1;
                        __hip_atomic_store(p.status, 16u + (unsigned)ph, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (16 + phase: a collect)
                    }
                    break;
                }
                if (spins > 64u) __builtin_amdgcn_s_sleep(1);
            }
            SM_STAMP(6);                            // 6: publish -> every awaited granule seen
            if (act[0]) {
                const f32x2 hv = {__builtin_bit_cast(float, val0[0]), __builtin_bit_cast(float, val1[0])};
                if (rg == 0) *reinterpret_cast<f32x2*>(h0buf + ((ph + 1) & 1) * H + 2 * pr0) = hv;
                // layer 1's input rows of this thread's group: the one value under each row's mask
#pragma unroll
                for (int k = 0; k < RG; ++k) {
                    const int r = rg + NG * k;
                    if (r < rv) *reinterpret_cast<f32x2*>(anext + r * SA + 2 * pr0) = f32x2{hv[0] * m0[k][0], hv[1] * m0[k][1]};
                }
            }
#pragma unroll
            for (int i = 0; i < NI1; ++i) {
                if (!act[1 + i]) continue;
                const f32x2 hv = {__builtin_bit_cast(float, val0[1 + i]), __builtin_bit_cast(float, val1[1 + i])};
                const int l = it_l[i], pr = it_pr[i], r = it_r[i];
                *reinterpret_cast<f32x2*>(anext + ((l - 1) * RC + r) * SA + H + 2 * pr) = hv;
                if (L > 2 && l < L - 1) *reinterpret_cast<f32x2*>(anext + (l * RC + r) * SA + 2 * pr) = f32x2{hv[0] * m1[i][0], hv[1] * m1[i][1]};
            }
        }
        SM_STAMP(7);                                // 7: values into LDS
        __syncthreads();                            // the next phase's operands are in LDS
        SM_STAMP(9);                                // 9: end-of-phase barrier
        if (ctl[0] != 0) return;
    }

    // ---- head: member m finishes row m of the cluster from the top layer's h(T-1) in its B tile; 16 lanes per target
    if (member < rv) {
        const int hw_o = tid >> 4, hw_part = tid & 15;
        float s_acc = 0.0f;
        if (hw_o < O) {
            const float* hv = a + (((P & 1) * LM + (LM - 1)) * RC + member) * SA + H + hw_part * (H / 16);
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
        if (hw_o < O && hw_part == 0) p.y[(size_t)(row0 + member) * O + hw_o] = s_acc + p.b_out[hw_o];
    }
    SM_STAMP(8);                                    // 8: head
#ifdef APE_CLUSTER_STAMPS
    if (p.dbg_wg != nullptr && tid == 0 && member == 0 && cluster == 0) {
        for (int k = 0; k < 10; ++k) p.dbg_wg[k] = st_acc[k];
        p.dbg_wg[10] = __builtin_amdgcn_s_memtime() - st_begin;
        p.dbg_wg[11] = __builtin_amdgcn_s_memrealtime() - st_rt0;
    }
#endif
    }   // rv > 0
    // ---- departure: the last workgroup out of the whole launch (idle clusters included) bumps the launch number and re-zeroes
    //      the XCD words (self-cleaning: a captured launch replays correctly)
    __syncthreads();
    if (tid == 0)
        ctl[2] = (__hip_atomic_fetch_add(p.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u) ? 1 : 0;
    __syncthreads();
    if (ctl[2] != 0) {
        // the next launch's tags differ from every tag of this one; when the 20-bit launch number wraps, the granules go back to
        // zero (tag 0 is never awaited)
        if (seq == 0xFFFFFu)
            for (size_t i = tid; i < (size_t)8 * p.gx_cluster_bytes / 4; i += 256)
                __hip_atomic_store(reinterpret_cast<unsigned*>(p.gx) + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid == 0) __hip_atomic_store(p.seq, seq + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(p.xcc_slots + 192 + tid, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid == 0) __hip_atomic_store(p.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// LDS: B tiles, layer 0's state, x slab, partial sums, control words, then the mask table (Philox: bits for up to 64 steps; injected:
// float multipliers of two phases)
template <int H, int L, int KX, int RC, bool INJ>
constexpr size_t mcs_smem() {
    return ((size_t)2 * (L - 1) * RC * (2 * H + 8) + 2 * H + 2 * (KX + 8) + (size_t)(L - 1) * (RC / 4) * 4 * 32 * 4 + 8) * sizeof(float) +
           (INJ ? (size_t)2 * (L - 1) * RC * H * sizeof(float) : (size_t)64 * (L - 1) * H * sizeof(unsigned short));
}

template <int H, int L, int KX, int RC, bool INJ, bool FED>
hipError_t launch_mcs(const McSmallParams& p, hipStream_t stream) {
    // (the keep / drop bits of the window's T steps, not of the 64 the table may hold: at the deployed T = 6 / 8 a member then needs
    //  little enough LDS for the feature builder's workgroup to share a CU with it)
    const size_t smem_bytes = mcs_smem<H, L, KX, RC, INJ>() - (INJ ? 0 : (size_t)(64 - p.T) * (L - 1) * H * sizeof(unsigned short));
    hipLaunchKernelGGL((ape_lstm_mc_small<H, L, KX, RC, INJ, FED>), dim3(8 * (H / 8) + (FED ? p.n_streams : 0)), dim3(256), smem_bytes, stream, p);
    return hipGetLastError();
}

template <int H, int L, int KX, bool INJ, bool FED>
hipError_t launch_mcs_rc(const McSmallParams& p, hipStream_t stream) {
    if (p.R <= 4) return launch_mcs<H, L, KX, 4, INJ, FED>(p, stream);
    if (p.R <= 8) return launch_mcs<H, L, KX, 8, INJ, FED>(p, stream);
    if (p.R <= 16) return launch_mcs<H, L, KX, 16, INJ, FED>(p, stream);
    return hipErrorInvalidValue;
}

template <int H, int L, int KX, int RC, bool INJ, bool FED>
hipError_t prepare_mcs_one() {
    // (what the instantiation asks for, not the CU's 160 KiB: the kernel also has static LDS -- the builder's row)
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&ape_lstm_mc_small<H, L, KX, RC, INJ, FED>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)mcs_smem<H, L, KX, RC, INJ>());
}
template <int H, int L, int KX, int RC>
hipError_t prepare_mcs_rc() {
    hipError_t e = prepare_mcs_one<H, L, KX, RC, false, false>();
    if (e == hipSuccess) e = prepare_mcs_one<H, L, KX, RC, true, false>();
    if (e == hipSuccess) e = prepare_mcs_one<H, L, KX, RC, false, true>();
    return e;
}
template <int H, int L, int KX>
hipError_t prepare_mcs() {
    hipError_t e = prepare_mcs_rc<H, L, KX, 4>();
    if (e == hipSuccess) e = prepare_mcs_rc<H, L, KX, 8>();
    if (e == hipSuccess) e = prepare_mcs_rc<H, L, KX, 16>();
    return e;
}

}  // namespace

bool ape_mc_small_supported(int H, int L, int KX) { return (H == 256 && L == 2 && KX == 32) || (H == 128 && L == 3 && KX == 64); }

// bytes of one cluster's granule region (two parities, 16 rows) -- the launch uses eight of them
size_t ape_mc_small_cluster_bytes(int H, int L) { return (size_t)2 * (H / 2) * (1 + (L - 1) * 16) * 16; }

hipError_t ape_prepare_lstm_mc_small(int H, int L, int KX) {
    if (H == 256 && L == 2 && KX == 32) return prepare_mcs<256, 2, 32>();
    if (H == 128 && L == 3 && KX == 64) return prepare_mcs<128, 3, 64>();
    return hipErrorInvalidValue;
}

// p.R rows per cluster (<= 16), p.cps clusters per stream, p.n_streams * p.cps <= 8, p.T <= 64; p.raw_rows: the frame's raw rows (host
// frames: the feature builder's workgroups ride along; Philox masks only)
template <int H, int L, int KX>
hipError_t launch_mcs_sel(const McSmallParams& p, hipStream_t stream) {
    const bool inj = (p.flags & APE_FLAG_DROPOUT_MASKS) != 0, fed = p.raw_rows != nullptr;
    if (inj && fed) return hipErrorInvalidValue;
    if (inj) return launch_mcs_rc<H, L, KX, true, false>(p, stream);
    return fed ? launch_mcs_rc<H, L, KX, false, true>(p, stream) : launch_mcs_rc<H, L, KX, false, false>(p, stream);
}
hipError_t ape_launch_lstm_mc_small(int H, int L, int KX, const McSmallParams& p, hipStream_t stream) {
    if (p.T > 64) return hipErrorInvalidValue;
    if (H == 256 && L == 2 && KX == 32) return launch_mcs_sel<256, 2, 32>(p, stream);
    if (H == 128 && L == 3 && KX == 64) return launch_mcs_sel<128, 3, 64>(p, stream);
    return hipErrorInvalidValue;
}
