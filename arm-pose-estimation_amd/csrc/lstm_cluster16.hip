// Weight-stationary cluster LSTM kernel, second generation, for the shapes whose members own 16 hidden units: the third deployed
// regressor (WatchPhoneUarmNN: I = 38, H = 128, L = 3; reference estimate/watch_phone_uarm_nn.py:13-41, nn_models.py:160-189),
// exact float32, eval mode, last-step output.
//
// lstm_cluster32.hip needs 32 hidden units per member (a wave = 8 units x 4 gates = the 32 columns of a 32x32x2 tile), i.e. four
// members per cluster at H = 128 -- half the chip at 1024 windows.  Here a member owns 16 units, a unit group of 4 of them = 16 columns
// ordered unit * 4 + gate on v_mfma_f32_16x16x4_f32; clusters of 8 members x 32 windows (RT = 2 row tiles of 16; 32 clusters = 256 CUs at
// 1024 windows); a workgroup = 4 unit groups x RT row tiles = eight waves, two per SIMD, ONE accumulator chain each (a single dependent
// chain already runs at the issue rate, tools/experiments/mfma_chain_rate.hip) -- the decomposition of the first-generation kernel
// (lstm_cluster.hip), with the second generation's exchange:
//   * exchange layout = LDS layout = fragment order [member][unit group][window][4 units]: a lane (window n, k-group g) reads the 16
//     bytes of member q's unit group g for its window -- units 16 q + 4 g + j, j = 0..3 -- and feeds four MFMAs (the weights are packed
//     with the same k permutation: the first generation's register image, wcl); gathered slices never pass through registers, a
//     layer-step's slice set (GH x 1 KB x RT) is copied global -> LDS by LDS-DMA, prefetched by the section in front;
//   * layers software-pipelined (phase p: layer l on step p - l): with three layers every section's hand-over has two other
//     sections to hide in;
//   * the workgroup barrier sits BETWEEN a section's two spans (input part, recurrent part); the cell update and the publish of a section
//     ride in single-MFMA slots of the NEXT section's first span (see `section` below);
//   * XCD-class clusters (arrival tickets within blockIdx % 8, verified at run time: plain stores in one XCD's L2, else
//     write-through), asynchronous flag look (inline-asm load, judged blocks later), bounded spins, sticky status, self-cleaning
//     -- all as lstm_cluster32.hip;
//   * the four gates of a (unit, window) cell land in one lane's four accumulator registers: lane-local cell update; the fresh
//     h value (one per lane) is transposed through a wave-private LDS patch into 16-byte pieces for the publish.
// Inline-asm MFMAs: hipcc guards neither their operands nor their results (tools/check_mfma_hazards.py checks the assembly): weights that
// do not fit the VGPR budget are AGPR operands explicitly, and a section's last MFMA carries the drain in its own asm statement.
#include <type_traits>

#include "ape_internal.h"
#include "async_look.h"
#include "../../include/ape_hip.h"

namespace {

typedef unsigned u32x4 __attribute__((__vector_size__(4 * sizeof(unsigned))));
constexpr unsigned SPIN_LIMIT = 1u << 22;

__device__ __forceinline__ float sigm(float v) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v)); }
__device__ __forceinline__ float tanh_(float v) { return 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.885390081777927f * v)) - 1.0f; }

// v_mfma_f32_16x16x4_f32: A = a weight register (row lane & 15 = unit * 4 + gate), B = an activation (column lane & 15 = window of
// the row tile), k = lane >> 4.  hipcc does not model an asm MFMA's result hazard: accumulators are read only behind mfma16_last().
// (AG: the weight lives in an accumulation register, read by the matrix core directly.  Never let the compiler park an operand of
//  these there by itself: its v_accvgpr_read right in front of the MFMA is a VALU write -> SrcA read without the wait states the
//  hardware needs, which hipcc cannot insert around inline asm -- seen as the first row tile of the top layer off by 2e-3.)
template <bool AG>
__device__ __forceinline__ void mfma16(f32x4& acc, float w, float a) {
    if constexpr (AG) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "a"(w), "v"(a));
    else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "v"(a));
}
// the last MFMA of a span that ends a section, with the drain in the SAME statement: the compiler, which takes an asm's result for ready,
// puts the phi copies of a branch merge right behind the MFMA otherwise (seen: the second row tile one k-step short, 1e-4)
template <bool AG>
__device__ __forceinline__ void mfma16_last(f32x4& acc, float w, float a) {
    if constexpr (AG) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0\n\ts_nop 15" : "+v"(acc) : "a"(w), "v"(a));
    else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0\n\ts_nop 15" : "+v"(acc) : "v"(w), "v"(a));
}

// NB k-blocks of 16: acc += W (registers w[w0 + 4 kb + j]) x activations (LDS: block kb at src + kb * stride, this lane's 16 bytes),
// fragments fetched one block ahead; mid(kb) behind block kb, fine(4 kb + j) behind every single MFMA
template <int NB, bool AG, bool DR, int NW, typename Mid, typename Fine>
__device__ __forceinline__ void span16(f32x4& acc, const float* __restrict__ src, int stride, const float (&w)[NW], int w0, Mid&& mid, Fine&& fine) {
    f32x4 a0 = *reinterpret_cast<const f32x4*>(src);
    f32x4 a1 = a0;
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
        if (kb + 1 < NB) a1 = *reinterpret_cast<const f32x4*>(src + stride * (kb + 1));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (DR && kb == NB - 1 && j == 3) mfma16_last<AG>(acc, w[w0 + 4 * kb + j], a0[j]);
            else {
                mfma16<AG>(acc, w[w0 + 4 * kb + j], a0[j]);
                fine(4 * kb + j);
            }
        }
        mid(kb);
        a0 = a1;
    }
}

__device__ __forceinline__ void dma_1k(unsigned lds_addr, unsigned voff, u32x4 rsrc, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen sc1 lds"
                 :: "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
template <bool WT>
__device__ __forceinline__ void store_16(u32x4 v, unsigned voff, u32x4 rsrc) {
    if constexpr (WT) asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen sc1" APE_STORE_TAIL :: "v"(v), "v"(voff), "s"(rsrc) : "memory");
    else asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen" APE_STORE_TAIL :: "v"(v), "v"(voff), "s"(rsrc) : "memory");
}
// (the flag looks: LDS-DMA into the wave's landing zone, async_look.h -- no destination register)

template <int H, int L, int KX, int RT>
__global__ __launch_bounds__(256 * RT, 3 - RT) void ape_lstm_cluster16(const ClusterParams p) {
    constexpr int GH = H / 16;              // members per cluster (16 units each, 4 per wave)
    constexpr int MR = 16 * RT;             // windows per cluster: RT tiles of 16 (RT = 2: one workgroup of eight waves per CU; RT = 1: two workgroups
                                            // of four waves per CU, each its own cluster member with its own barrier)
    constexpr int NWV = 4 * RT, NT = 64 * NWV;
    constexpr int BX = KX / 16, BH = H / 16;// k-blocks of 16 (one block = one member's units, or 16 input columns)
    constexpr int NWX = 4 * BX, NWH = 4 * BH;               // weight registers per lane: input part of layer 0 / an H-wide part
    constexpr int NW0 = NWX + NWH, NWU = 2 * NWH;           // layer 0 / a layer above
    constexpr int NFL = NWV * GH;           // flags per (cluster, layer): one per member wave
    constexpr int BLK = 4 * MR * 4;         // floats of one k-block in LDS: [wave / k-group 4][window 32][4] = 2 KB
    constexpr int HL = GH * BLK;            // floats of one slice set
    constexpr int XL = BX * BLK;            // floats of the x slab (the same fragment order)
    constexpr int NDMA = HL * 4 / 1024 / NWV; // LDS-DMA instructions per wave and gather
    constexpr unsigned SET_BYTES = HL * sizeof(float);
    static_assert(L == 3 && NFL <= 64 && (GH & (GH - 1)) == 0 && NDMA >= 1, "shape");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef APE_CLUSTER_STAMPS
    const unsigned long long tl_entry = __builtin_amdgcn_s_memrealtime();
#endif
    // eight waves, two per SIMD: unit group ug (4 hidden units = 16 tile columns) x row tile rt (16 windows).  A wave cannot issue anything
    // while its own MFMA occupies the matrix core (tools/experiments/mfma_chain_rate.hip: one v_fma between two MFMAs costs 12 cycles), so
    // the waves (ug, 0) and (ug, 1), which hold the same weights, take turns there: one's cell update, exchange and operand fetches run
    // under the other's matrix work
    const int ug = wave & 3, rt = wave >> 2;
    const int n = lane & 15, g = lane >> 4;         // window of a row tile; k-group of the operands = hidden unit of the results
    const int T = p.T, I = p.I, O = p.O;
    const bool normalize = (p.flags & APE_FLAG_NORMALIZE_INPUT) != 0;
    const bool bcast_x = (p.flags & APE_FLAG_BROADCAST_X) != 0;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    // h of layer l < L-1: two parity buffers (read by the layer's recurrence AND as the next layer's input); top layer: one
    float* hbase = smem;                                  // [2 (L-1) + 1][HL]
    float* xin = hbase + (2 * (L - 1) + 1) * HL;          // [XL]
    float* patch = xin + XL;                              // [wave 8][window 16][4]: the publish transpose
    f32x4* bias_s = reinterpret_cast<f32x4*>(patch + 4 * MR * 4);     // [wave 4][L][g 4]: start values of unit g's four gates
    unsigned* look_s = reinterpret_cast<unsigned*>(bias_s + 4 * L * 4);   // [wave NWV][64]: landing zones of the flag looks (async_look.h)
    int* ctl = reinterpret_cast<int*>(look_s + NWV * 64);             // [0] abort, [1] class ticket, [2] last-out, [3] same XCD
    unsigned* arrive = reinterpret_cast<unsigned*>(ctl + 4);          // [L]: publishes of layer l whose stores have drained, all waves of this member
    auto hb = [&](int l, int par) -> float* { return hbase + (l < L - 1 ? 2 * l + par : 2 * (L - 1)) * HL; };

    unsigned* const class_ticket = p.xcc_slots + 64;
    unsigned* const xcc_words = p.xcc_slots + 64 + 8 * 16;
    const int cls = blockIdx.x & 7;
    unsigned my_xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(my_xcc));
    my_xcc &= 0xFu;
    if (tid < L) arrive[tid] = 0u;
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
    const int row0 = cluster * MR;
    if (tid == 0)
        __hip_atomic_store(xcc_words + cluster * GH + member, 0x10u | my_xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    // ---- x: thread -> NE (window, column) elements of the step slab, all with the same column ---------------------------------
    constexpr int NE = (MR * KX) / NT;
    const int xk = tid % KX, xrow = tid / KX;
    const int rows_here = bcast_x ? MR : max(0, min(MR, p.B - row0));
    const unsigned long long x_addr = reinterpret_cast<unsigned long long>(p.x + (bcast_x ? (size_t)0 : (size_t)row0 * T * I));
    const unsigned x_lo = __builtin_amdgcn_readfirstlane((unsigned)x_addr), x_hi = __builtin_amdgcn_readfirstlane((unsigned)(x_addr >> 32));
    const int x_bytes = __builtin_amdgcn_readfirstlane((int)((size_t)(bcast_x ? 1 : rows_here) * T * I * sizeof(float)));
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<float*>(((unsigned long long)x_hi << 32) | x_lo), 0, x_bytes, 0x00020000);
    const unsigned x_rowbytes = bcast_x ? 0u : (unsigned)(T * I * sizeof(float));
    const unsigned x_off0 = (xk < I) ? (unsigned)xrow * x_rowbytes + (unsigned)(xk * sizeof(float)) : 0x80000000u;
    const unsigned x_estride = (unsigned)(NT / KX) * x_rowbytes;
    float xr[NE];
    auto fetch_x = [&](int t) {
        const int slot = (t + p.x_ring >= T) ? t + p.x_ring - T : t + p.x_ring;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const unsigned off = (xrow + e * (NT / KX) < rows_here) ? x_off0 + (unsigned)e * x_estride : 0x80000000u;
            xr[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rsrc, off, (unsigned)(slot * I * sizeof(float)), 0));
        }
    };
    const double x_mean = (normalize && xk < I) ? p.xx_m[xk] : 0.0;
    const double x_std = (normalize && xk < I) ? p.xx_s[xk] : 1.0;
    const double x_rstd = (normalize && xk < I) ? p.xx_r[xk] : 1.0;
    // column k of window w lives at [k-block k / 16][k-group (k % 16) / 4][window w][k % 4]
    const int x_slot = ((xk >> 4) * 4 + ((xk & 15) >> 2)) * (MR * 4) + (xk & 3);
    // (x - mean) / std in float64 like the reference (numpy), rounded to float32 once: q1 is the correctly rounded quotient
    auto stage_x1 = [&](int e) {
        float v = xr[e];
        if (normalize) {
            const double d = (double)xr[e] - x_mean;
            const double q0 = d * x_rstd;
            const double rr = fma(-q0, x_std, d);
            const double q1 = fma(rr, x_rstd, q0);
            v = (float)((rr == rr) ? q1 : q0);
        }
        xin[x_slot + (xrow + e * (NT / KX)) * 4] = v;
    };
    auto stage_x = [&]() {
#pragma unroll
        for (int e = 0; e < NE; ++e) stage_x1(e);
    };
    fetch_x(0);

    // ---- weights: registers for the whole launch; host layout of the first generation (ape_api.hip, wcl):
    //      [member][wave][register / 4][lane][4], register 4 q + j of lane (row c = lane & 15 = unit * 4 + gate, k-group g) =
    //      [W_ih | W_hh][gate * H + member * 16 + wave * 4 + unit][16 q + 4 g + j]
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
    if (tid < 4 * L * 4) {                  // start values (b_ih + b_hh): [wave][layer][unit g] -> the four gates
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
    unsigned* const flags_of = p.xflags + (size_t)cluster * L * NFL;
    const ape_desc_t fl_desc = ape_make_desc(p.xflags, (unsigned)((gridDim.x / GH) * L * NFL * sizeof(unsigned)));
    const unsigned fl_off = (unsigned)(cluster * L * NFL * sizeof(unsigned));
    const unsigned look_voff = (unsigned)((lane & (NFL - 1)) * sizeof(unsigned));
    const unsigned look_lds = (unsigned)reinterpret_cast<unsigned long long>(look_s) + (unsigned)(wave * 256);
    const unsigned* const look_mine = look_s + wave * 64 + lane;
    auto hx_base = [&](int l, int par) -> unsigned { return (unsigned)((((size_t)cluster * L + l) * 2 + par) * SET_BYTES); };
    const unsigned hbase_lds = (unsigned)reinterpret_cast<unsigned long long>(hbase);

    stage_x();
    if (T > 1) fetch_x(1);

    if (wave == 0) {                        // do all members of this cluster share an XCD?
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
        // hand-over form (ape_internal.h): write-through (`sc1`) payload stores unless the caller opted into the plain in-XCD form AND the
    // members were verified to share an XCD; uniform over the cluster
    const bool in_l2 = APE_HANDOVER_IN_L2(p.flags, ctl[3] != 0);

    auto wait_flags = [&](int l, unsigned want) {
        unsigned spins = 0;
        while (true) {
            unsigned v = want;
            if (lane < NFL) v = __hip_atomic_load(flags_of + l * NFL + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
    auto opaque = [](unsigned v) -> unsigned { asm volatile("" : "+s"(v)); return v; };
    const unsigned dma_voff = (unsigned)(lane * 16);
    const unsigned wave_kib = (unsigned)(wave * 1024);
    // slices of layer l, step `step` -> its LDS buffer; wave w copies KiB w, w + 4, ...
    auto issue_piece = [&](int l, int step, int k) {
        const unsigned src = hx_base(l, step & 1) + wave_kib + (unsigned)(k * NWV * 1024);
        const unsigned buf = (unsigned)(l < L - 1 ? 2 * l + (step & 1) : 2 * (L - 1));
        dma_1k(opaque(hbase_lds + wave_kib) + buf * SET_BYTES + (unsigned)(k * NWV * 1024), dma_voff, hx_desc, src);
    };
    // The flag of a publish is raised per MEMBER, by the last of its waves whose store has drained (an arrival counter per layer in LDS), as
    // ONE store instruction over the member's NWV flag words: 64 waves each storing its own word of the same two cache lines at about the
    // same time serialise on the memory side -- write-through stores to one line complete one after the other -- and the last of them became
    // visible 3 .. 7 us after it was issued (round 6, `profiles/r06_flag_serialisation.md`: the final gather of this kernel waited that long).
    // No wave can wait for a flag that needs its own arrival's siblings stuck behind it: a blocking wait (`wait_flags`, in front of a section's
    // barrier) is for a slice set published a phase ago, whose raise -- in front of the barrier of the section after its publish at the
    // latest -- every wave of the member has passed; the final gather's waves all reach their raise without a barrier in between.
    int pend_idx = -1;                      // layer of the pending publish
    unsigned pend_epoch = 0u;
    auto raise_pending = [&]() {
        if (pend_idx < 0) return;
        unsigned prev = 0u;
        if (lane == 0) prev = __hip_atomic_fetch_add(arrive + pend_idx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        prev = __builtin_amdgcn_readfirstlane(prev);
        if (prev + 1u == (unsigned)NWV * pend_epoch && lane < NWV)
            __hip_atomic_store(flags_of + pend_idx * NFL + member * NWV + lane, pend_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pend_idx = -1;
    };
    auto bar = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    // Section (ph, l) = layer l on step t = ph - l, in two spans with the workgroup barrier BETWEEN them:
    //   span A  x_t / h^{l-1}_t: in LDS since before the barrier of the section in front (layer l-1 gathered h^{l-1}_t as ITS recurrent
    //           input; x was staged two sections ago), so it starts without waiting for anybody -- and carries, in its first blocks,
    //           the cell update and the publish of the section in front, whose accumulators were only drained at its end: the gate
    //           arithmetic (20 transcendentals per lane) and the exchange store run under this section's matrix work;
    //   ------  the slice set prefetched by the section in front (h^l_{t-1}) has landed: counted wait, barrier;
    //   span B  h^l_{t-1}; in its blocks: flag of the store above, the look at the next section's flags, its gather.
    // Vector-memory queue of a wave at the barrier of a steady-state section: [x fetch, section 1 only] [gather DMA (NDMA)] | publish
    // store (1) -- `vmcnt(1)`.
    const int P = T + L - 1;
    bool prefetched = false;
    const int frag = (g * MR + rt * 16 + n) * 4;                  // this lane's 16 bytes inside a k-block
    float* const my_patch = patch + wave * 64;
#ifdef APE_CLUSTER_STAMPS
    unsigned long long dg[L][5] = {};                             // steady-state sections: cycles in span A, counted wait, barrier, span B; count
    auto now = [&]() -> unsigned long long {
        const unsigned long long c = __builtin_readcyclecounter();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        return c;
    };
#endif
#ifdef APE_CLUSTER_STAMPS
    // timeline of cluster 0 / member 0 / wave 0 (100 MHz counter): [0] kernel entry, [1] prologue done, [2 + 3 ph + l] end of section (ph, l),
    // then final publish, final gather, head, exit (tests/tools/timeline_uarm16.py)
    unsigned long long tl[64];
    int tl_n = 0;
    auto tl_mark = [&]() { if (tl_n < 64) tl[tl_n++] = __builtin_amdgcn_s_memrealtime(); };
    tl[tl_n++] = tl_entry;
    tl_mark();
#endif
    f32x4 pa = f32x4{0.0f, 0.0f, 0.0f, 0.0f};                    // drained accumulator of the section in front
    auto section = [&](auto steady_tag, auto layer_tag, const int ph) -> bool {
        constexpr bool ST = decltype(steady_tag)::value;
        constexpr int l = decltype(layer_tag)::value;
        const int t = ph - l;
        const bool active = ST || (t >= 0 && t < T);
        const bool need = ST || (t >= 1 && t <= T);
        // the section in front: layer lp on step tp
        constexpr int lp = (l + L - 1) % L;
        const int tp = (l == 0) ? t - L : t + 1;
        const bool pactive = ST || (tp >= 0 && tp < T);
        // the next section: layer ln on step tn; the slice set it is missing is h^{ln}_{tn-1}, epoch tn
        constexpr int ln = (l + 1 < L) ? l + 1 : 0;
        const int tn = (l + 1 < L) ? t - 1 : t + L;
        const bool pre_ok = ST || (tn >= 1 && tn <= T);
        unsigned peek = (unsigned)tn;
        bool go = false;
        float hn = 0.0f;
        // the publish: transpose through the wave's LDS patch ([window 16][4 units]); lanes 0..15 send one window's 16 bytes each
        // (exactly ONE store instruction per wave and section: the counted wait at the barrier relies on it)
        auto publish = [&]() {
            const f32x4 hf = *reinterpret_cast<const f32x4*>(my_patch + (lane & 15) * 4);     // (same wave: LDS operations are in order)
            // (whole-vector cast: hipcc 7.2 folds a per-element cast of a loaded vector into a one-dword load + splat)
            const u32x4 hv = __builtin_bit_cast(u32x4, hf);
            const unsigned off = (pactive && lane < 16) ? hx_base(lp, tp & 1) + (unsigned)(((member * 4 + ug) * MR + rt * 16 + lane) * 16) : 0x80000000u;
            if (in_l2) store_16<false>(hv, off, hx_desc);
            else store_16<true>(hv, off, hx_desc);
            if (pactive) {
                pend_idx = lp;
                pend_epoch = (unsigned)(tp + 1);
            }
        };
        // ---- work in front of the barrier (fill and drain sections, in one piece): cell update of the section in front, its publish
        auto pre = [&]() {
            if (pactive) {
                // registers 0..3 = i, f, g, o of unit g, window rt * 16 + n
                const float iv = sigm(pa[0]), fv = sigm(pa[1]), gv = tanh_(pa[2]), ov = sigm(pa[3]);
                const float c = fv * cst[lp] + iv * gv;
                cst[lp] = c;
                hn = ov * tanh_(c);
            }
            my_patch[n * 4 + g] = hn;
            publish();
        };
        // the same work in the steady state, one slot behind every MFMA of span A: slots 0 .. 9 the cell update (four exponentials, four
        // reciprocals, the cell, its hyperbolic tangent), slot 10 the patch, slot 13 the store; every result is pinned by an empty volatile
        // asm so that the compiler leaves the step in its slot
        float ge[4], gc = 0.0f;
        auto pin = [](float& v) { asm volatile("" : "+v"(v)); };
        auto gate_slot = [&](int i) {
            if (i < 4) {
                ge[i] = __builtin_amdgcn_exp2f((i == 2 ? -2.885390081777927f : -1.4426950408889634f) * pa[i]);
                pin(ge[i]);
            } else if (i < 8) {
                const float r = __builtin_amdgcn_rcpf(1.0f + ge[i - 4]);
                ge[i - 4] = (i == 6) ? 2.0f * r - 1.0f : r;
                pin(ge[i - 4]);
            } else if (i == 8) {
                const float c = ge[1] * cst[lp] + ge[0] * ge[2];
                cst[lp] = c;
                gc = __builtin_amdgcn_exp2f(-2.885390081777927f * c);
                pin(gc);
            } else if (i == 9) {
                hn = ge[3] * (2.0f * __builtin_amdgcn_rcpf(1.0f + gc) - 1.0f);
                pin(hn);
            } else if (i == 10) {
                my_patch[n * 4 + g] = hn;
            } else if (i == 13) {
                publish();
            }
        };
        // ---- work behind the barrier, item k: 0 flag of the store above, x; 1 look at the next section's flags; 2 judge;
        //      3 .. 2 + NDMA the next section's gather.  (The look lands in the wave's LDS zone, async_look.h: rounds 3-4 had it land in a
        //      register that the compiler "must not copy in between" -- nothing could have stopped it.)
        auto post = [&](int k) {
            if (k == 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the publish store has drained
                raise_pending();
                if constexpr (l == 1 && !ST) {
                    // x of the next layer-0 step: registers -> LDS (its readers, span A of this phase's layer-0 section, finished
                    // before this section's barrier; the next ones start behind the barrier of section 2); steady state: in slots
                    if (ph + 1 < T) stage_x();
                }
            } else if (k == 1) {
                look_issue(look_lds, look_voff, fl_desc, fl_off + (unsigned)(ln * NFL * sizeof(unsigned)));     // (always: no branch around it)
            } else if (k == 2) {
                look_landed();
                peek = *look_mine;
                go = pre_ok && __all((int)(peek >= (unsigned)tn)) != 0;
                if constexpr (l == 1) {
                    // the fetch of the step after it: the oldest entries of the memory queue when the next counted wait comes
                    if (ST || ph + 2 < T) fetch_x(ph + 2);
                }
            } else if (go) {
                issue_piece(ln, tn - 1, k - 3);
            }
        };
        constexpr int QF = 0, QP = 1, QJ = 4;
        static_assert(QJ + NDMA <= BH && QF < QP && QP < QJ && BX >= 4, "hook schedule");
#ifdef APE_CLUSTER_STAMPS
        const unsigned long long c0 = ST ? now() : 0ull;
#endif
        f32x4 acc;
        auto hook_a = [&](int q) { if (!ST && q == 0) pre(); };
#ifdef APE_ABL_NOGATES
        auto fine_a = [&](int i) { if (ST && i == 13) gate_slot(i); };
#else
        auto fine_a = [&](int i) { if (ST && i <= 13) gate_slot(i); };
#endif
        if (active) {
            acc = bias_s[(ug * L + l) * 4 + g];
            if constexpr (l == 0) span16<BX, false, !ST, NW0>(acc, xin + frag, BLK, w0, 0, hook_a, fine_a);
            else span16<BH, true, !ST, NWU>(acc, hb(l - 1, t & 1) + frag, BLK, wu[l - 1], 0, hook_a, fine_a);
        } else {
            pre();
        }
#ifdef APE_CLUSTER_STAMPS
        const unsigned long long c1 = ST ? now() : 0ull;
#endif
        // ---- this layer's slices of its last step into LDS ------------------------------------------------------------------------
        if (need) {
            if (!prefetched) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                raise_pending();
                wait_flags(l, (unsigned)t);
#pragma unroll
                for (int k = 0; k < NDMA; ++k) issue_piece(l, t - 1, k);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            }
        }
        prefetched = false;
#ifdef APE_CLUSTER_STAMPS
        const unsigned long long c2 = ST ? now() : 0ull;
#endif
        bar();
#ifdef APE_CLUSTER_STAMPS
        const unsigned long long c3 = ST ? now() : 0ull;
#endif
        const int abort_word = ctl[0];
        if (active && (ST || t > 0)) {
            auto hook = [&](int q) {
                if (q == QF) post(0);
                if (q == QP) post(1);
                if (q >= QJ && q < QJ + NDMA) {
                    if (q == QJ) post(2);
                    post(3 + q - QJ);
                }
            };
            // (steady state, section 1: one element of the next x slab every other slot of blocks 1 .. 2, behind the store's drain)
            auto fine_b = [&](int i) {
                if constexpr (ST && l == 1) {
                    if (i >= 4 && i < 4 + 2 * NE && ((i - 4) & 1) == 0) stage_x1((i - 4) >> 1);
                }
            };
            if constexpr (l == 0) span16<BH, false, true, NW0>(acc, hb(0, (t - 1) & 1) + frag, BLK, w0, NWX, hook, fine_b);
            else span16<BH, true, true, NWU>(acc, hb(l, (t - 1) & 1) + frag, BLK, wu[l - 1], NWH, hook, fine_b);
        } else {
#pragma unroll
            for (int k = 0; k < 3 + NDMA; ++k) post(k);
        }
        if (active) pa = acc;
#ifdef APE_CLUSTER_STAMPS
        if (ST) {
            const unsigned long long c4 = now();
            dg[l][0] += c1 - c0; dg[l][1] += c2 - c1; dg[l][2] += c3 - c2; dg[l][3] += c4 - c3; dg[l][4] += 1;
        }
#endif
        if (abort_word != 0) return false;
        if (go) prefetched = true;
        return true;
    };
    bool ok = true;
#pragma unroll 1
    for (int ph = 0; ph < P && ok; ++ph) {
        const bool st = ph >= L && ph <= T - 3;
        ok = st ? section(std::true_type{}, std::integral_constant<int, 0>{}, ph) : section(std::false_type{}, std::integral_constant<int, 0>{}, ph);
#ifdef APE_CLUSTER_STAMPS
        tl_mark();
#endif
        if (!ok) break;
        ok = st ? section(std::true_type{}, std::integral_constant<int, 1>{}, ph) : section(std::false_type{}, std::integral_constant<int, 1>{}, ph);
#ifdef APE_CLUSTER_STAMPS
        tl_mark();
#endif
        if (!ok) break;
        ok = st ? section(std::true_type{}, std::integral_constant<int, 2>{}, ph) : section(std::false_type{}, std::integral_constant<int, 2>{}, ph);
#ifdef APE_CLUSTER_STAMPS
        tl_mark();
#endif
    }
    if (!ok) return;
#ifdef APE_CLUSTER_STAMPS
    if (p.dbg_wg != nullptr && lane == 0 && cluster == 0 && member == 0)
        for (int l = 0; l < L; ++l)
            for (int k = 0; k < 5; ++k) p.dbg_wg[wave * 16 + l * 5 + k] = dg[l][k];
#endif
    // ---- the last section's cell update and publish (layer L-1, step T-1) ----------------------------------------------------------------
    {
        const float iv = sigm(pa[0]), fv = sigm(pa[1]), gv = tanh_(pa[2]), ov = sigm(pa[3]);
        my_patch[n * 4 + g] = ov * tanh_(fv * cst[L - 1] + iv * gv);
        const f32x4 hf = *reinterpret_cast<const f32x4*>(my_patch + (lane & 15) * 4);
        const u32x4 hv = __builtin_bit_cast(u32x4, hf);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        raise_pending();
        const unsigned off = (lane < 16) ? hx_base(L - 1, (T - 1) & 1) + (unsigned)(((member * 4 + ug) * MR + rt * 16 + lane) * 16) : 0x80000000u;
        if (in_l2) store_16<false>(hv, off, hx_desc);
        else store_16<true>(hv, off, hx_desc);
        pend_idx = L - 1;
        pend_epoch = (unsigned)T;
    }
#ifdef APE_CLUSTER_STAMPS
    tl_mark();
#endif
    // ---- final gather: h^{L-1}_{T-1} of every member ----------------------------------------------------------------------------
    {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        raise_pending();
#ifdef APE_CLUSTER_STAMPS
        tl_mark();
        if (p.dbg_wg != nullptr && lane == 0 && cluster == 0) {       // when did every member wave of cluster 0 raise its last flag, and enter?
            p.dbg_wg[256 + member * 8 + wave] = __builtin_amdgcn_s_memrealtime();
            if (wave == 0) p.dbg_wg[320 + member] = tl_entry;
        }
#endif
        if (!prefetched) {
            wait_flags(L - 1, (unsigned)T);
#ifdef APE_CLUSTER_STAMPS
            tl_mark();
#endif
#pragma unroll
            for (int k = 0; k < NDMA; ++k) issue_piece(L - 1, T - 1, k);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef APE_CLUSTER_STAMPS
        tl_mark();
#endif
        bar();
        if (ctl[0] != 0) return;
    }
#ifdef APE_CLUSTER_STAMPS
    tl_mark();
#endif
    // ---- head: member m finishes windows (MR / GH) m .. of the cluster's 32; 4 lanes per (window, target) ----------------------------
    {
        constexpr int RPM = MR / GH;
        const float* htop = hb(L - 1, 0);
        const int part = tid & 3;
        for (int oi = tid >> 2; oi < ((RPM * O + NT / 4 - 1) / (NT / 4)) * (NT / 4); oi += NT / 4) {
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
    if (p.dbg_wg != nullptr && tid == 0 && cluster == 0 && member == 0) {
        p.dbg_wg[128] = (unsigned long long)tl_n;
        for (int i = 0; i < tl_n; ++i) p.dbg_wg[129 + i] = tl[i];
    }
#endif
    // ---- self-cleaning -----------------------------------------------------------------------------------------------------------------
    __syncthreads();
    if (tid == 0)
        ctl[2] = (__hip_atomic_fetch_add(p.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) ? 1 : 0;
    __syncthreads();
    if (ctl[2] != 0) {
        const int n_flags = (int)(gridDim.x / GH) * L * NFL;
        for (int i = tid; i < n_flags; i += NT) __hip_atomic_store(p.xflags + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int i = tid; i < (int)gridDim.x; i += NT) __hip_atomic_store(xcc_words + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid < 8) __hip_atomic_store(class_ticket + tid * 16, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid == 0) __hip_atomic_store(p.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <int H, int L, int KX, int RT>
constexpr size_t smem16() {
    return ((size_t)(2 * (L - 1) + 1) * (H / 16) * 256 * RT + (size_t)(KX / 16) * 256 * RT + 4 * RT * 64) * sizeof(float) + (size_t)4 * L * 4 * 16 + (size_t)4 * RT * 64 * sizeof(unsigned) + 16 + 16;
}

}  // namespace

bool ape_cluster16_supported(int H, int L, int KX) { return H == 128 && L == 3 && KX == 64; }

hipError_t ape_prepare_lstm_cluster16(int H, int L, int KX) {
    if (!ape_cluster16_supported(H, L, KX)) return hipSuccess;
    static_assert(smem16<128, 3, 64, 2>() <= APE_LDS_BYTES && 2 * smem16<128, 3, 64, 1>() <= APE_LDS_BYTES, "LDS layout exceeds a CU");
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ape_lstm_cluster16<128, 3, 64, 2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       APE_LDS_BYTES);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&ape_lstm_cluster16<128, 3, 64, 1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               APE_LDS_BYTES / 2);
}

// `rows` windows (at most 1024 on a whole MI355X); the grid is rounded up to whole block-index classes (8 clusters x GH members).
// Default form: 32-window clusters, one eight-wave workgroup per CU.  APE_FLAG_ALT_FORM (include/ape_hip.h): 16-window clusters,
// two four-wave workgroups per CU, each with its own barrier -- built to see whether two cluster members that drift freely on a CU overlap
// better than two row tiles behind one barrier: they do not (1024 x 64 on one box: 456 vs 450 us)
hipError_t ape_launch_lstm_cluster16(int H, int L, int KX, int rows, const ClusterParams& p, hipStream_t stream) {
    if (!ape_cluster16_supported(H, L, KX)) return hipErrorInvalidValue;
    if (!(p.flags & APE_FLAG_ALT_FORM)) {
        const int grid_clusters = ((rows + 31) / 32 + 7) / 8 * 8;
        constexpr size_t smem = smem16<128, 3, 64, 2>();
        hipLaunchKernelGGL((ape_lstm_cluster16<128, 3, 64, 2>), dim3(grid_clusters * 8), dim3(512), smem, stream, p);
    } else {
        const int grid_clusters = ((rows + 15) / 16 + 7) / 8 * 8;
        constexpr size_t smem = smem16<128, 3, 64, 1>();
        hipLaunchKernelGGL((ape_lstm_cluster16<128, 3, 64, 1>), dim3(grid_clusters * 8), dim3(256), smem, stream, p);
    }
    return hipGetLastError();
}
