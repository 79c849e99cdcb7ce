// Weight-stationary cluster LSTM kernel for the calls of the 3 x 128 upper-arm regressor that fit ONE launch -- its deployed short window
// first (WatchPhoneUarmNN: I = 38, H = 128, L = 3, sequence_len 6; reference estimate/watch_phone_uarm_nn.py:13-41,107-121,
// nn_models.py:160-189) --, exact float32, eval mode, last-step output.  Round 6.
//
// lstm_cluster.hip / lstm_cluster16.hip software-pipeline the three layers (phase p: layer l on step p - l) and hide a section's hand-over
// behind the two other sections of the phase.  At T = 6 that pipeline is eight phases of which four are fill / drain: their hand-overs
// have nothing to hide behind, the idle sections still pay their barriers and waits, and both row tiles of a CU wait at the same time
// (lstm_cluster16.hip's timeline at 1024 x 6: 58 us in the kernel for 26 us of matrix work, profiles/r06_uarm_T6_timelines.md).  This kernel turns
// the decomposition around:
//   * LEVEL-synchronous: level k = every (layer l, step k - l) that exists, computed back to back from the slices of level k - 1 --
//     ONE hand-over per level (T + 2 of them), no idle sections, no hooks inside the MFMA spans, every wait a plain blocking one;
//   * clusters of 8 members x 32 windows, a workgroup of EIGHT waves = two AGENTS of four (one row tile of 16 windows each; wave = unit
//     group of 4 hidden units = 16 tile columns on v_mfma_f32_16x16x4_f32, the first generation's register image `wcl`: 48 VGPRs + 128 AGPRs
//     of weights; wave w and w + 4 share a SIMD).  The agents are independent clusters' worth of work that ALTERNATE on the matrix cores by
//     construction: a wave starts a level only when its SIMD partner has finished one (a word per wave in LDS) -- agent 0's level k, agent
//     1's level k, agent 0's level k + 1 .. -- so one computes while the other one's slices travel.  (Two free-running workgroups per CU
//     were measured first: the two drift into phase, compute at the same time at half speed each and then wait at the same time -- 72 us.)
//     The agents never meet at s_barrier (it has no subsets on gfx950): an agent's barrier is an arrival counter in LDS;
//   * hand-over by TAGGED GRANULES, no flags: a wave publishes its 64 fresh values as 64 {value, tag} pairs of 8 bytes (one write-through
//     store per lane, tag = launch number and level), and the consumers poll the granules themselves -- 16-byte loads, four or eight behind
//     one wait -- until every tag is this level's, then write the values to LDS in fragment order [member][unit group][window 16][4 units].
//     One memory hop instead of four (store acknowledged, flag stored, flag seen, slices gathered: that form of this kernel took 63 us,
//     this one 54); double-buffered by level parity on both sides.  The only atomicity assumed is a naturally aligned 8-byte store's;
//   * valid under ANY placement (write-through stores, L1-bypassing loads); the clusters still form within block-index classes
//     (blockIdx % 8 = XCD under round-robin dispatch) for speed, nothing relies on it, there is no plain-store variant and no rendezvous;
//   * a row tile past the end of the batch computes nothing and waits for nobody; up to 512 rows the launcher asks for ONE row tile per
//     cluster (APE_FLAG_LV16_SINGLE: agent 1 of every workgroup idle) so that the rows spread over every CU -- nobody alternates then, a level
//     costs its matrix work plus its hand-over, and that still beats the first generation's pipeline from 5 rows on at every window length;
//   * bounded spins (every wait, LDS ones included, ends when anybody in the workgroup has given up), sticky status word; nothing to clean
//     but the class tickets: the launch number goes up by one, the next launch awaits other tags.
// Measured at 1024 x 6: 54-55 us against the first generation's 61.5, at 512 x 6 40.0 against 42.8; a level costs the two agents' matrix work plus
// ~1 us -- beside a partner's dependent MFMA chain the other wave's compares and LDS writes hardly issue, so the end of a hand-over waits
// for the partner's level to end (sub-stamps in the diagnostic build).
// Inline asm: the MFMAs (weights of the upper layers are AGPR operands; a level-section's last MFMA carries its drain) and the polling
// loads (their wait inside the statement) -- statement forms of lstm_cluster16.hip / lstm_latency_common.h, scanned by
// tools/check_mfma_hazards.py; the publish store is the compiler's own (its data hazards are the compiler's to pad).
#include <type_traits>

#include "ape_internal.h"
#include "async_look.h"
#include "../../include/ape_hip.h"

namespace {

typedef unsigned u32x4 __attribute__((__vector_size__(4 * sizeof(unsigned))));
constexpr unsigned SPIN_LIMIT = 1u << 22;

__device__ __forceinline__ float sigm(float v) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v)); }
__device__ __forceinline__ float tanh_(float v) { return 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.885390081777927f * v)) - 1.0f; }

// v_mfma_f32_16x16x4_f32: A = a weight register (row lane & 15 = unit * 4 + gate), B = an activation (column lane & 15 = window),
// k = lane >> 4.  AG: the weight lives in an accumulation register, named as such ("a") -- never parked there by the compiler
// (its v_accvgpr_read in front of an asm MFMA has no wait states: lstm_cluster16.hip)
template <bool AG>
__device__ __forceinline__ void mfma16(f32x4& acc, float w, float a) {
    if constexpr (AG) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "a"(w), "v"(a));
    else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "v"(a));
}
// the last MFMA of a level-section, with the drain in the SAME statement (hipcc takes an asm's result for ready)
template <bool AG>
__device__ __forceinline__ void mfma16_last(f32x4& acc, float w, float a) {
    if constexpr (AG) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0\n\ts_nop 15" : "+v"(acc) : "a"(w), "v"(a));
    else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0\n\ts_nop 15" : "+v"(acc) : "v"(w), "v"(a));
}

// NB k-blocks of 16: acc += W (registers w[w0 + 4 kb + j]) x activations (LDS: block kb at src + kb * stride, this lane's 16 bytes),
// fragments fetched one block ahead; DR: the span ends the section -- its last MFMA drains.  (DR is a template argument and the caller
// branches AROUND whole spans: a branch merge between the last MFMA and its drain is where hipcc puts phi copies of the accumulator.)
// NB k-blocks of 16: acc += W (registers w[w0 + 4 kb + j]) x activations (LDS: block kb at src + kb * stride, this lane's 16 bytes),
// fragments fetched one block ahead; DR: the span ends the section -- its last MFMA drains.  (DR is a template argument and the caller
// branches AROUND whole spans: a branch merge between the last MFMA and its drain is where hipcc puts phi copies of the accumulator.)
// (Measured and dropped, round 6: the first fragment of every span fetched under the span in front -- 55.6 against 54.8 us at 1024 x 6, the
//  LDS latency at a span's start is covered by the SIMD partner; two or four interleaved accumulator chains -- 55.0 / 56.8 us: the partner's
//  hand-over instructions issue sooner between independent MFMAs, the matrix work itself gets slower by as much.)
template <int NB, bool AG, bool DR, int NW>
__device__ __forceinline__ void span16(f32x4& acc, const float* __restrict__ src, int stride, const float (&w)[NW], int w0) {
    f32x4 a0 = *reinterpret_cast<const f32x4*>(src);
    f32x4 a1 = a0;
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
        if (kb + 1 < NB) a1 = *reinterpret_cast<const f32x4*>(src + stride * (kb + 1));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (DR && kb == NB - 1 && j == 3) mfma16_last<AG>(acc, w[w0 + 4 * kb + j], a0[j]);
            else mfma16<AG>(acc, w[w0 + 4 * kb + j], a0[j]);
        }
        a0 = a1;
    }
}

// NI polling loads (16 bytes = two {value, tag} granules per lane, L1-bypassing) and the wait for them in ONE statement: the compiler knows
// nothing of the asynchronous return, so no use (or copy) of a result may be scheduled between a load and the wait (lstm_latency_common.h;
// the leading s_nop: a descriptor reloaded from a spill lane right in front needs its five wait states)
__device__ __forceinline__ void poll_pairs4(u32x4 (&v)[4], const unsigned (&off)[4], u32x4 rsrc, unsigned soff) {
    asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %4, %8, %9 offen sc1\n\tbuffer_load_dwordx4 %1, %5, %8, %9 offen sc1\n\t"
                 "buffer_load_dwordx4 %2, %6, %8, %9 offen sc1\n\tbuffer_load_dwordx4 %3, %7, %8, %9 offen sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3])
                 : "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]), "s"(rsrc), "s"(soff) : "memory");
}


// the same over TWO slice sets at once (uniform offsets soff0 / soff1, the lanes' offsets shared): eight loads behind one wait -- a level's
// sweep costs round trips, not bytes
__device__ __forceinline__ void poll_pairs8(u32x4 (&v)[4], u32x4 (&u)[4], const unsigned (&off)[4], u32x4 rsrc, unsigned soff0, unsigned soff1) {
    asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %8, %12, %13 offen sc1\n\tbuffer_load_dwordx4 %1, %9, %12, %13 offen sc1\n\t"
                 "buffer_load_dwordx4 %2, %10, %12, %13 offen sc1\n\tbuffer_load_dwordx4 %3, %11, %12, %13 offen sc1\n\t"
                 "buffer_load_dwordx4 %4, %8, %12, %14 offen sc1\n\tbuffer_load_dwordx4 %5, %9, %12, %14 offen sc1\n\t"
                 "buffer_load_dwordx4 %6, %10, %12, %14 offen sc1\n\tbuffer_load_dwordx4 %7, %11, %12, %14 offen sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(u[0]), "=&v"(u[1]), "=&v"(u[2]), "=&v"(u[3])
                 : "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]), "s"(rsrc), "s"(soff0), "s"(soff1) : "memory");
}

template <int H, int L, int KX>
__global__ __launch_bounds__(512, 1) void ape_lstm_level16(const ClusterParams p) {
    constexpr int GH = H / 16;              // members per cluster (16 units each, 4 per wave)
    constexpr int MR = 16;                  // windows per ROW TILE; a cluster owns two (32 windows), one per agent
    constexpr int NA = 2;                   // agents per workgroup: waves 0..3 and 4..7, wave w and w + 4 share a SIMD
    constexpr int NTA = 256;                // threads per agent
    constexpr int BX = KX / 16, BH = H / 16;// k-blocks of 16 (one block = one member's units, or 16 input columns)
    constexpr int NWX = 4 * BX, NWH = 4 * BH;
    constexpr int NW0 = NWX + NWH, NWU = 2 * NWH;
    constexpr int BLK = 4 * MR * 4;         // floats of one k-block in LDS: [k-group 4][window 16][4] = 1 KB
    constexpr int HL = GH * BLK;            // floats of one slice set (8 KB)
    constexpr int XL = BX * BLK;            // floats of the x slab (the same fragment order)
    constexpr unsigned GSET_BYTES = HL * 8;  // one slice set in the exchange buffer: {value, tag} granules (16 KB)
    constexpr int AGENT_FLOATS = 2 * L * HL + 2 * XL;                // an agent's LDS: slice sets, x slabs
    static_assert(L == 3 && GH == 8 && HL == 8 * NTA, "built for the 3 x 128 model: four granule pairs per thread and slice set");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int agent = wave >> 2, ug = wave & 3;     // row tile; unit group: hidden units 16 member + 4 ug ..
    const int tid_a = tid & (NTA - 1);
#ifdef APE_CLUSTER_STAMPS
    const unsigned long long tl_entry = __builtin_amdgcn_s_memrealtime();
#endif
    const int n = lane & 15, g = lane >> 4;         // window; k-group of the operands = hidden unit of the results
    const int T = p.T, I = p.I, O = p.O;
    const bool normalize = (p.flags & APE_FLAG_NORMALIZE_INPUT) != 0;
    const bool bcast_x = (p.flags & APE_FLAG_BROADCAST_X) != 0;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* hset = smem + agent * AGENT_FLOATS;            // [2 level parity][L][HL]: the slice sets gathered behind level k live in parity k & 1
    float* xin = hset + 2 * L * HL;                       // [2 step parity][XL]
    f32x4* bias_s = reinterpret_cast<f32x4*>(smem + NA * AGENT_FLOATS);   // [unit group 4][L][g 4]: start values of unit g's four gates (both agents)
    int* ctl = reinterpret_cast<int*>(bias_s + 4 * L * 4);            // [0] abort, [1] class ticket, [2] last-out
    unsigned* bar_cnt = reinterpret_cast<unsigned*>(ctl + 4);         // [NA]: arrivals at the agent's barrier (4 per generation)
    unsigned* fin = bar_cnt + 4;                                      // [NA][4]: levels whose matrix work wave (agent, ug) has finished

    unsigned* const class_ticket = p.xcc_slots + 64;
    const int cls = blockIdx.x & 7;
    if (tid < 16) bar_cnt[tid] = 0u;                      // (bar_cnt[0..3], fin[0..7], 4 spare words)
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
    // one-agent form (the launcher's choice for up to 16 x max_clusters rows): a cluster owns ONE row tile, agent 1 of every workgroup is idle,
    // so that 512 rows still spread over every CU (32 clusters) -- with nobody to alternate with, a level costs its matrix work plus its
    // hand-over, which is still less than the first generation's pipeline at these sizes
    const bool single = (p.flags & APE_FLAG_LV16_SINGLE) != 0;
    const int row0 = single ? cluster * MR : cluster * (NA * MR) + agent * MR;

    // ---- x: thread -> NE (window, column) elements of the agent's step slab, all with the same column ----------------------------------
    constexpr int NE = (MR * KX) / NTA;
    const int xk = tid_a % KX, xrow = tid_a / KX;
    const int rows_here = (single && agent != 0) ? 0 : (bcast_x ? MR : max(0, min(MR, p.B - row0)));
    const unsigned long long x_addr = reinterpret_cast<unsigned long long>(p.x + (bcast_x ? (size_t)0 : (size_t)row0 * T * I));
    // (per-agent descriptor: its words differ between the two halves of the workgroup, uniform within a wave)
    const unsigned x_lo = __builtin_amdgcn_readfirstlane((unsigned)x_addr), x_hi = __builtin_amdgcn_readfirstlane((unsigned)(x_addr >> 32));
    const int x_bytes = __builtin_amdgcn_readfirstlane((int)((size_t)(bcast_x ? 1 : rows_here) * T * I * sizeof(float)));
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<float*>(((unsigned long long)x_hi << 32) | x_lo), 0, x_bytes, 0x00020000);
    const unsigned x_rowbytes = bcast_x ? 0u : (unsigned)(T * I * sizeof(float));
    const unsigned x_off0 = (xk < I) ? (unsigned)xrow * x_rowbytes + (unsigned)(xk * sizeof(float)) : 0x80000000u;
    const unsigned x_estride = (unsigned)(NTA / KX) * x_rowbytes;
    float xr[NE];
    auto fetch_x = [&](int t) {
        const int slot = (t + p.x_ring >= T) ? t + p.x_ring - T : t + p.x_ring;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const unsigned off = (xrow + e * (NTA / KX) < rows_here) ? x_off0 + (unsigned)e * x_estride : 0x80000000u;
            xr[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rsrc, off, (unsigned)(slot * I * sizeof(float)), 0));
        }
    };
    const double x_mean = (normalize && xk < I) ? p.xx_m[xk] : 0.0;
    const double x_std = (normalize && xk < I) ? p.xx_s[xk] : 1.0;
    const double x_rstd = (normalize && xk < I) ? p.xx_r[xk] : 1.0;
    // column k of window w lives at [k-block k / 16][k-group (k % 16) / 4][window w][k % 4]
    const int x_slot = ((xk >> 4) * 4 + ((xk & 15) >> 2)) * (MR * 4) + (xk & 3);
    // (x - mean) / std in float64 like the reference (numpy), rounded to float32 once: q1 is the correctly rounded quotient
    // (staging every slab of a short window in the prologue was measured and lost: 58.0 against 54.8 us at 1024 x 6 -- a hand-over lasts as
    //  long as the SIMD partner's matrix work whatever it contains, and the prologue grew by 1.5 us)
    auto stage_x = [&](int t) {
        float* dst = xin + (t & 1) * XL;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            float v = xr[e];
            if (normalize) {
                const double d = (double)xr[e] - x_mean;
                const double q0 = d * x_rstd;
                const double rr = fma(-q0, x_std, d);
                const double q1 = fma(rr, x_rstd, q0);
                v = (float)((rr == rr) ? q1 : q0);
            }
            dst[x_slot + (xrow + e * (NTA / KX)) * 4] = v;
        }
    };
    fetch_x(0);
    // the launch number of this model's level kernel (bumped by the last workgroup out): the upper bits of every tag.  Loaded in FRONT of the
    // weights: the vector-memory counter is in issue order, so a value loaded behind them would be waited for behind all 44 weight loads --
    // and level 0 needs only the first four of those
    const unsigned seq = __hip_atomic_load(p.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0xFFFFFu;

    // ---- weights: registers for the whole launch (both agents hold the member's slices); host layout of the first generation
    //      (ape_api.hip, wcl): [member][wave][register / 4][lane][4], register 4 q + j of lane (row c = lane & 15 = unit * 4 + gate,
    //      k-group g) = [W_ih | W_hh][gate * H + member * 16 + wave * 4 + unit][16 q + 4 g + j]
    float w0[NW0];
    float wu[L - 1][NWU];
    {
        const f32x4* s0 = reinterpret_cast<const f32x4*>(p.wcl[0]) + ((size_t)(member * 4 + ug) * (NW0 / 4)) * 64 + lane;
#pragma unroll
        for (int i = 0; i < NW0 / 4; ++i) {
            const f32x4 v = s0[i * 64];
            w0[4 * i] = v[0]; w0[4 * i + 1] = v[1]; w0[4 * i + 2] = v[2]; w0[4 * i + 3] = v[3];
        }
#pragma unroll
        for (int l = 1; l < L; ++l) {
            const f32x4* s1 = reinterpret_cast<const f32x4*>(p.wcl[l]) + ((size_t)(member * 4 + ug) * (NWU / 4)) * 64 + lane;
#pragma unroll
            for (int i = 0; i < NWU / 4; ++i) {
                const f32x4 v = s1[i * 64];
                wu[l - 1][4 * i] = v[0]; wu[l - 1][4 * i + 1] = v[1]; wu[l - 1][4 * i + 2] = v[2]; wu[l - 1][4 * i + 3] = v[3];
            }
        }
    }
    if (tid < 4 * L * 4) {                  // start values (b_ih + b_hh): [unit group][layer][unit g] -> the four gates
        const int wv = tid / (L * 4), l = (tid / 4) % L, gg = tid & 3;
        f32x4 bv;
#pragma unroll
        for (int gate = 0; gate < 4; ++gate) bv[gate] = p.bias[l][gate * H + member * 16 + wv * 4 + gg];
        bias_s[tid] = bv;
    }
    float cst[L];
#pragma unroll
    for (int l = 0; l < L; ++l) cst[l] = 0.0f;

    const unsigned long long hx_addr = reinterpret_cast<unsigned long long>(p.hx);
    u32x4 hx_desc;
    hx_desc[0] = __builtin_amdgcn_readfirstlane((unsigned)hx_addr);
    hx_desc[1] = __builtin_amdgcn_readfirstlane((unsigned)(hx_addr >> 32) & 0xFFFFu);
    hx_desc[2] = __builtin_amdgcn_readfirstlane((unsigned)p.hx_bytes);
    hx_desc[3] = 0x00020000u;
    // exchange: [cluster][agent 2][level parity 2][layer L][member][unit group][lane 64] granules {h of (unit lane >> 4, window lane & 15), tag}:
    // a wave publishes its 64 fresh values as 64 granules (one 8-byte write-through store per lane); nobody raises or reads a flag -- the
    // consumers poll the granules themselves until every tag names this launch and this level (one hop instead of store-ack, flag, look, gather)
    const int tile = cluster * NA + agent;
    auto gx_base = [&](int par, int l) -> unsigned { return (unsigned)((((size_t)tile * 2 + par) * L + l) * GSET_BYTES); };
    const __amdgpu_buffer_rsrc_t gx_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.hx, 0, (int)p.hx_bytes, 0x00020000);
    const unsigned pub_off = (unsigned)(((member * 4 + ug) * 64 + lane) * 8);
    // sweep: thread tid_a takes pairs tid_a + 256 q (q = 0..3) of a set = granules 2 tid_a + 512 q (+ 1): producer member 2 q + (tid_a >> 7),
    // unit group (tid_a >> 5) & 3, unit (tid_a & 31) >> 3, windows 2 (tid_a & 7) (+ 1) -> LDS [member][unit group][window][unit]
    unsigned sw_off[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) sw_off[q] = (unsigned)((tid_a + 256 * q) * 16);
    const int sw_lds = (tid_a >> 7) * BLK + ((tid_a >> 5) & 3) * 64 + 2 * (tid_a & 7) * 4 + ((tid_a & 31) >> 3);

    // ---- synchronisation inside the workgroup: the two agents never meet at s_barrier (it has no subsets on gfx950) ----------------------
    // a wait on an LDS word: bounded, and it ends when anybody in the workgroup has given up (ctl[0])
    auto spin_ge = [&](const unsigned* word, unsigned want) {
        unsigned spins = 0;
        while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < want) {
            if (__hip_atomic_load(ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0) return;
            if (++spins > SPIN_LIMIT) {
                if (lane == 0) {
                    __hip_atomic_store(ctl, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_store(p.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                return;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    };
    // barrier over the agent's four waves: an arrival counter in LDS (this wave's LDS traffic has completed in front of the arrival)
    unsigned bar_gen = 0u;
    auto abar = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(bar_cnt + agent, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        bar_gen += 4u;
        spin_ge(bar_cnt + agent, bar_gen);
    };

    stage_x(0);
    if (T > 1) fetch_x(1);
#ifdef APE_CLUSTER_STAMPS
    unsigned long long tl[64];
    unsigned sp_cnt[16];
    unsigned long long sub[8] = {};
    int tl_n = 0;
    auto tl_mark = [&]() { if (tl_n < 64) tl[tl_n++] = __builtin_amdgcn_s_memrealtime(); };
    tl[tl_n++] = tl_entry;
#endif
    __syncthreads();                        // bias_s, x_0 (the last time both agents meet before the end)
#ifdef APE_CLUSTER_STAMPS
    tl_mark();
#endif

    const int frag = (g * MR + n) * 4;                            // this lane's 16 bytes inside a k-block
    unsigned* const my_fin = fin + agent * 4 + ug;
    const unsigned* const peer_fin = fin + (agent ^ 1) * 4 + ug;  // the wave this one shares its SIMD with
    const int P = T + L - 1;
    // a row tile past the end of the batch (all eight members of the tile agree): nothing to compute, nothing to hand over -- its waves
    // only tell their SIMD partners that the matrix core is theirs for the whole launch
    const bool tile_live = rows_here > 0;
    if (!tile_live && lane == 0) __hip_atomic_store(my_fin, 0x7FFFFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll 1
    for (int k = 0; k < (tile_live ? P : 0); ++k) {
        // the matrix core of this SIMD is the other agent's while it works on a level: agent 0 goes first (level k after the peer's
        // level k - 1), agent 1 second (after the peer's level k) -- one computes while the other one's slices travel, by construction
        spin_ge(peer_fin, (unsigned)(k + 1 - (agent ^ 1)));
#ifdef APE_CLUSTER_STAMPS
        tl_mark();
#endif
        const float* const hprev = hset + ((k - 1) & 1) * L * HL;     // the slices of level k - 1
        const unsigned tag = (seq << 12) | (unsigned)(k + 1);
        // ---- every (layer l, step k - l) that exists, back to back ------------------------------------------------------------------
        auto layer = [&](auto layer_tag) {
            constexpr int l = decltype(layer_tag)::value;
            const int t = k - l;
            if (t < 0 || t >= T) return;                          // uniform over the grid
            f32x4 acc = bias_s[(ug * L + l) * 4 + g];
            // (step 0: h_{-1} = 0, no recurrent span)
            if constexpr (l == 0) {
                if (t > 0) {
                    span16<BX, false, false, NW0>(acc, xin + (t & 1) * XL + frag, BLK, w0, 0);
                    span16<BH, false, true, NW0>(acc, hprev + frag, BLK, w0, NWX);
                } else {
                    span16<BX, false, true, NW0>(acc, xin + (t & 1) * XL + frag, BLK, w0, 0);
                }
            } else {
                if (t > 0) {
                    span16<BH, true, false, NWU>(acc, hprev + (l - 1) * HL + frag, BLK, wu[l - 1], 0);
                    span16<BH, true, true, NWU>(acc, hprev + l * HL + frag, BLK, wu[l - 1], NWH);
                } else {
                    span16<BH, true, true, NWU>(acc, hprev + (l - 1) * HL + frag, BLK, wu[l - 1], 0);
                }
            }
            // registers 0..3 = i, f, g, o of unit g, window n
            const float iv = sigm(acc[0]), fv = sigm(acc[1]), gv = tanh_(acc[2]), ov = sigm(acc[3]);
            const float c = fv * cst[l] + iv * gv;
            cst[l] = c;
            const float hnew = ov * tanh_(c);
            // the publish: this lane's value as ONE granule {value, tag}, write-through
            typedef unsigned u32x2 __attribute__((__vector_size__(2 * sizeof(unsigned))));
            const u32x2 gr = {__builtin_bit_cast(unsigned, hnew), tag};
            __builtin_amdgcn_raw_buffer_store_b64(gr, gx_rsrc, pub_off, gx_base(k & 1, l), 16 /* sc1: write-through */);
        };
        layer(std::integral_constant<int, 0>{});
        layer(std::integral_constant<int, 1>{});
        layer(std::integral_constant<int, 2>{});
        // (handing the SIMD over one section early -- the peer's barrier and its look at this word under this level's last section -- was measured
        //  and changed nothing: 55.7 us)
        if (lane == 0) __hip_atomic_store(my_fin, (unsigned)(k + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // the SIMD is the peer's
        // (What a hand-over costs beside a computing partner, measured with sub-stamps: the polling loads 0.3-0.5 us as alone, but a set's 8
        //  compares 0.7 us and its 4 LDS writes 1.0 us against 0.2 / 0.1 alone -- the partner's DEPENDENT MFMA chain leaves other waves of the
        //  SIMD hardly an issue slot; a hand-over ends when the partner's matrix work does, whatever it contains.  s_setprio 3 around the
        //  hand-over changed nothing: it is not the arbiter's choice.)
#ifdef APE_CLUSTER_STAMPS
        tl_mark();
#endif
        // the next step's x: registers -> LDS (its parity's last readers finished a level ago)
        if (k + 1 < T) stage_x(k + 1);
        if (k + 2 < T) fetch_x(k + 2);
#ifdef APE_CLUSTER_STAMPS
        tl_mark();
#endif
#ifdef APE_CLUSTER_STAMPS
        if (k == 4) sub[5] = __builtin_amdgcn_s_memrealtime();
#endif
        // ---- hand-over: sweep the granules of the sets the next level (or the head) reads -- the layers active in this level, in the order
        //      their producers publish them -- until every tag is this level's; values -> LDS in fragment order
        {
            // the layers whose sets are read next: a contiguous range [lo, hi] (the head: the top layer only)
            int lo = (k >= T) ? k - T + 1 : 0, hi = (k < L - 1) ? k : L - 1;
            if (k == P - 1) lo = L - 1;
            auto fresh4 = [&](const u32x4 (&w)[4]) -> bool {
                return w[0][1] == tag && w[0][3] == tag && w[1][1] == tag && w[1][3] == tag &&
                       w[2][1] == tag && w[2][3] == tag && w[3][1] == tag && w[3][3] == tag;
            };
            // (the values go to LDS as the words they are: hipcc 7.2 folds a per-element bit cast of a vector into a splat of element 0 --
            //  seen here as `ds_write2_b32 .., v62, v62`, both windows of a pair taking the first one's value)
            auto commit = [&](const u32x4 (&w)[4], int l) {
                unsigned* dst = reinterpret_cast<unsigned*>(hset + ((k & 1) * L + l) * HL + sw_lds);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    dst[q * 2 * BLK] = w[q][0];
                    dst[q * 2 * BLK + 4] = w[q][2];
                }
            };
            unsigned spins = 0;
            auto give_up = [&]() -> bool {           // bounded: somebody in the workgroup has given up, or this wave does
                if (__hip_atomic_load(ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0) return true;
                if (++spins > SPIN_LIMIT || __hip_atomic_load(p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                    if (lane == 0) {
                        __hip_atomic_store(ctl, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_store(p.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    return true;
                }
                __builtin_amdgcn_s_sleep(1);
                return false;
            };
            u32x4 v[4];
            if (hi > lo) {                           // two sets behind one wait (their producers published them a section apart)
                u32x4 u[4];
                while (true) {
                    poll_pairs8(v, u, sw_off, hx_desc, gx_base(k & 1, lo), gx_base(k & 1, lo + 1));
#ifdef APE_CLUSTER_STAMPS
                    if (k == 4) sub[0] = __builtin_amdgcn_s_memrealtime();
#endif
                    if (__all((int)(fresh4(v) && fresh4(u)))) break;
                    if (give_up()) break;
                }
#ifdef APE_CLUSTER_STAMPS
                if (k == 4) sub[1] = __builtin_amdgcn_s_memrealtime();
#endif
                commit(v, lo);
                commit(u, lo + 1);
#ifdef APE_CLUSTER_STAMPS
                if (k == 4) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); sub[2] = __builtin_amdgcn_s_memrealtime(); }
#endif
                lo += 2;
            }
            if (lo <= hi) {
                while (true) {
                    poll_pairs4(v, sw_off, hx_desc, gx_base(k & 1, lo));
#ifdef APE_CLUSTER_STAMPS
                    if (k == 4) sub[3] = __builtin_amdgcn_s_memrealtime();
#endif
                    if (__all((int)fresh4(v))) break;
                    if (give_up()) break;
                }
                commit(v, lo);
#ifdef APE_CLUSTER_STAMPS
                if (k == 4) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); sub[4] = __builtin_amdgcn_s_memrealtime(); }
#endif
            }
#ifdef APE_CLUSTER_STAMPS
            if (k < 16) sp_cnt[k] = spins;
#endif
        }
#ifdef APE_CLUSTER_STAMPS
        tl_mark();
#endif
        abar();
#ifdef APE_CLUSTER_STAMPS
        tl_mark();
#endif
        if (__hip_atomic_load(ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0) return;
    }
    // ---- head: member m finishes windows (MR / GH) m .. of the agent's 16; 4 lanes per (window, target) --------------------------------
    if (tile_live) {
        constexpr int RPM = MR / GH;
        const float* htop = hset + (((P - 1) & 1) * L + (L - 1)) * HL;
        const int part = tid_a & 3;
        for (int oi = tid_a >> 2; oi < ((RPM * O + NTA / 4 - 1) / (NTA / 4)) * (NTA / 4); oi += NTA / 4) {
            const bool live = oi < RPM * O;
            const int rr = live ? oi / O : 0, o = live ? oi - rr * O : 0;
            const int row = member * RPM + rr, b = row0 + row;
            float s_acc = 0.0f;
            if (live) {
                const float* wv = p.w_out + (size_t)o * H;
                // units 4 q .. 4 q + 3 (q = k / 4) of window `row` live at htop[(q * MR + row) * 4]
                for (int q = part; q < H / 4; q += 4) {
                    const f32x4 a0 = *reinterpret_cast<const f32x4*>(htop + (q * MR + row) * 4);
                    const f32x4 u0 = *reinterpret_cast<const f32x4*>(wv + q * 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) s_acc = fmaf(a0[j], u0[j], s_acc);
                }
            }
            s_acc += __shfl_xor(s_acc, 1, 64);
            s_acc += __shfl_xor(s_acc, 2, 64);
            if (p.y != nullptr && live && part == 0 && b < p.B) p.y[(size_t)b * O + o] = s_acc + p.b_out[o];
        }
    }
#ifdef APE_CLUSTER_STAMPS
    tl_mark();
    if (p.dbg_wg != nullptr && tid_a == 0 && blockIdx.x < 256) {        // every agent's entry and exit (by block index), and its ticket
        p.dbg_wg[256 + 3 * (blockIdx.x * 2 + agent)] = tl_entry;
        p.dbg_wg[257 + 3 * (blockIdx.x * 2 + agent)] = tl[tl_n - 1];
        p.dbg_wg[258 + 3 * (blockIdx.x * 2 + agent)] = (unsigned long long)(tile * GH + member);
    }
    if (p.dbg_wg != nullptr && tid_a == 0 && cluster == 0 && member == 0) {
        for (int i = 0; i < 16 && i < P; ++i) p.dbg_wg[1800 + 16 * agent + i] = sp_cnt[i];
        for (int i = 0; i < 6; ++i) p.dbg_wg[1840 + 8 * agent + i] = sub[i];
        p.dbg_wg[128 - 64 * agent] = (unsigned long long)tl_n;
        for (int i = 0; i < tl_n && i < 63; ++i) p.dbg_wg[129 - 64 * agent + i] = tl[i];
    }
#endif
    // ---- self-cleaning (s_barrier does not count waves that have ended: an agent that left early does not hold the other up) ---------
    __syncthreads();
    if (tid == 0)
        ctl[2] = (__hip_atomic_fetch_add(p.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) ? 1 : 0;
    __syncthreads();
    if (ctl[2] != 0) {
        // the next launch's tags differ from every tag of this one; when the 20-bit launch number wraps, the granules go back to zero (tag 0
        // is never awaited) so that a tag of 2^20 launches ago cannot be taken for a fresh one
        if (seq == 0xFFFFFu)
            for (size_t i = tid; i < p.hx_bytes / 4; i += NA * NTA)
                __hip_atomic_store(reinterpret_cast<unsigned*>(p.hx) + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid == 0) __hip_atomic_store(p.seq, seq + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid < 8) __hip_atomic_store(class_ticket + tid * 16, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid == 0) __hip_atomic_store(p.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <int H, int L, int KX>
constexpr size_t smem_level16() {
    return (size_t)2 * ((size_t)2 * L * (H / 16) * 256 + (size_t)2 * (KX / 16) * 256) * sizeof(float) + (size_t)4 * L * 4 * 16 + 16 + 64;
}

}  // namespace

bool ape_level16_supported(int H, int L, int KX) { return H == 128 && L == 3 && KX == 64; }

// 32-window clusters (two row tiles of 16, one per agent) of 8 workgroups, one workgroup per CU, whole block-index classes of 8 clusters
int ape_level16_max_clusters(int n_cus) { return (n_cus / 8) / 8 * 8; }
// its exchange buffer: [cluster][agent 2][parity 2][layer 3] sets of 16 KB ({value, tag} granules); the model keeps it apart from the
// flag-based kernels' buffer (tags of an aborted launch are reset with it) behind a 256-byte header that holds the launch number
size_t ape_level16_gx_bytes(int n_cus) { return (size_t)ape_level16_max_clusters(n_cus) * 2 * 2 * 3 * 16 * 1024; }

hipError_t ape_prepare_lstm_level16(int H, int L, int KX) {
    if (!ape_level16_supported(H, L, KX)) return hipSuccess;
    static_assert(smem_level16<128, 3, 64>() <= APE_LDS_BYTES, "LDS layout exceeds a CU");
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&ape_lstm_level16<128, 3, 64>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)smem_level16<128, 3, 64>());
}

// `rows` windows, at most 32 x ape_level16_max_clusters(n_cus) (16 x with APE_FLAG_LV16_SINGLE in p.flags: one row tile per cluster); the grid is
// rounded up to whole block-index classes (8 clusters x 8 members)
hipError_t ape_launch_lstm_level16(int H, int L, int KX, int rows, const ClusterParams& p, hipStream_t stream) {
    if (!ape_level16_supported(H, L, KX)) return hipErrorInvalidValue;
    const bool single = (p.flags & APE_FLAG_LV16_SINGLE) != 0;
    const int grid_clusters = single ? ((rows + 15) / 16 + 7) / 8 * 8 : ((rows + 31) / 32 + 7) / 8 * 8;
    constexpr size_t smem = smem_level16<128, 3, 64>();
    hipLaunchKernelGGL((ape_lstm_level16<128, 3, 64>), dim3(grid_clusters * 8), dim3(512), smem, stream, p);
    return hipGetLastError();
}
