// fp16 cluster LSTM kernel, second generation (BASELINE.json configs[4]: watch-only model, 1024 windows x 64
// frames, "fp16 hidden state with fp32 accumulate").  Same arithmetic and storage precision as
// lstm_cluster_f16.hip (binary16 W, x, h; f32 accumulate, cell state and head; reference semantics
// estimate/nn_models.py:169-174,180-189), restructured around what bounds that kernel: the cluster-wide exchange of
// h, one fabric round trip per phase with nothing to hide it behind (its f16 MFMAs are ~10 % of a phase).
//
//   * 8 members per cluster instead of 16: a member owns 32 hidden units of every layer (wave: 8 units = two
//     16-column MFMA tiles, 200 weight registers per lane as binary16 pairs), a cluster owns 32 windows, 32 clusters
//     fill the 256 CUs.  Half the peers to wait for, half the flag fan-in, 16 KB instead of 64 KB gathered per
//     phase and member.
//   * the cluster's 32 windows are two ROW SETS of 16 that take turns: while set A's fresh h is on its way through
//     the exchange, the member computes set B, and vice versa.  The sets share nothing (windows are independent), so
//     unlike a layer pipeline there is no second dependent round trip per phase; all layers of a set are computed
//     back to back from the LDS state of its last phase (layer l works on step p - l) and published together.
//   * XCD-pure clusters where the dispatcher allows it: cluster membership is by arrival ticket WITHIN the block
//     index class (blockIdx % 8) -- under the placement observed on MI355X (workgroups dealt round-robin over the 8
//     XCDs) the 8 members of a cluster then share one XCD, hence one L2.  Nothing is assumed: every member
//     publishes the XCD it really runs on (HW_REG_XCC_ID) and only if all 8 agree does the cluster exchange with
//     plain stores (acknowledged by that L2, not written through to memory) and plain flag stores; otherwise the
//     any-placement form (sc1 write-through stores, agent-scope flag stores).  Loads of handed-off bytes are sc1
//     buffer loads either way (MI355X guide G16 / visibility table row 1: per-wave flag after the wave's own
//     vmcnt(0), consumer polls then loads).  A class has exactly gridDim/8 workgroups = whole clusters, so every
//     cluster forms once its workgroups are dispatched; spins are bounded and raise the sticky status word.
//   * self-cleaning like the other cluster kernels: the last workgroup out re-zeroes every polled word.
#include "ape_internal.h"
#include "../../include/ape_hip.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((__vector_size__(4 * sizeof(unsigned))));
constexpr unsigned SPIN_LIMIT = 1u << 22;

__device__ __forceinline__ float gate_act(float v, bool is_tanh) {
    const float e = __builtin_amdgcn_exp2f((is_tanh ? -2.885390081777927f : -1.4426950408889634f) * v);
    const float s = __builtin_amdgcn_rcpf(1.0f + e);
    return is_tanh ? 2.0f * s - 1.0f : s;
}

// v_mfma_f32_16x16x32_f16 with the A operand (a weight fragment that lives in AGPRs for the whole launch) pinned to
// the accumulator register file, so that the 256 architectural VGPRs stay free for a whole section's activation
// fragments (fetched up front: a single ds_read_b128 feeds only two of these 16-cycle MFMAs, so reads issued one
// block ahead leave the matrix pipe waiting on LDS latency -- 100 cycles per block instead of 32).  hipcc does not
// model an asm MFMA's result hazard: the accumulators are read only after mfma_drain().
__device__ __forceinline__ void mfma_wa(f32x4& acc, const f32x4 w, const f32x4 a) {
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(w), "v"(a));
}
// (the accumulators are in/out operands of the drain: the gate math, which reads them, cannot be scheduled above it)
template <int NTW>
__device__ __forceinline__ void mfma_drain(f32x4 (&acc)[NTW]) {
    static_assert(NTW == 2, "two tiles per wave");
    asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[1]));
}

// NQ 32-deep k-blocks of LDS activations (16 rows x 32 k each) into registers
template <int NQ>
__device__ __forceinline__ void load_frags(f32x4 (&a)[NQ], const _Float16* __restrict__ src) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) a[q] = *reinterpret_cast<const f32x4*>(src + 32 * q);
}
// acc[t] += W_t (registers) x A over NQ k-blocks
template <int NTW, int NQ, int NW>
__device__ __forceinline__ void span(f32x4 (&acc)[NTW], const f32x4 (&a)[NQ], const f32x4 (&w)[NTW][NW], int w_off) {
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int t = 0; t < NTW; ++t) mfma_wa(acc[t], w[t][w_off + q], a[q]);
}

// Diagnostic build (make diag: -DAPE_CLUSTER_STAMPS): shader-cycle sums per section kind of one workgroup's wave 0,
// written to the model's debug words (memory nothing else reads).  The shipped library has none of this code.
#ifdef APE_CLUSTER_STAMPS
#define V2_STAMP(k)                                                       \
    do {                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();     \
        st_acc[k] += now_ - st_t0;                                        \
        st_t0 = now_;                                                     \
        __builtin_amdgcn_sched_barrier(0);                                \
    } while (0)
#else
#define V2_STAMP(k) do {} while (0)
#endif

template <int H, int L, int KX>
__global__ __launch_bounds__(256, 1) void ape_lstm_cluster_f16v2(const ClusterParams p) {
    constexpr int UPW = 8;                  // hidden units per wave
    constexpr int NTW = UPW / 4;            // 16-column MFMA tiles per wave (column = unit * 4 + gate)
    constexpr int GH = H / (4 * UPW);       // members per cluster
    constexpr int SR = 16, NS = 2;          // rows per set, sets per cluster
    constexpr int SH = H + 16, SX = KX + 16; // LDS row strides in halves (16-byte multiples, conflict-free b128 reads)
    constexpr int QX = KX / 32, QH = H / 32;
    constexpr int NB0 = QX + QH, NB1 = 2 * QH;
    constexpr int NFL = 4 * GH;             // flags per (cluster, set): one per member wave
    constexpr int PIECES = L * GH * 4 * SR; // 16-byte pieces of one set's gathered slices
    constexpr int NGV = PIECES / 256;
    static_assert(GH == 8 && L == 2 && PIECES % 256 == 0, "built for 8-member clusters of the 2 x 256 models");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int T = p.T, I = p.I, O = p.O;
    const bool normalize = (p.flags & APE_FLAG_NORMALIZE_INPUT) != 0;
    const bool bcast_x = (p.flags & APE_FLAG_BROADCAST_X) != 0;
#if defined(APE_CLUSTER_STAMPS) || defined(APE_ABLATE)
    // timing-only ablations (outputs are wrong): what a part really costs = the launch time with and without it
    // (`make ablate` builds them without the stamps, whose own waits distort a kernel this short)
    const bool d_noex = (p.flags & APE_DIAG_NO_EXCHANGE) != 0, d_noact = (p.flags & APE_DIAG_NO_ACT) != 0;
    const bool d_nomfma = (p.flags & APE_DIAG_NO_MFMA) != 0, d_nox = (p.flags & APE_DIAG_NO_XSTAGE) != 0;
#else
    constexpr bool d_noex = false, d_noact = false, d_nomfma = false, d_nox = false;
#endif

    extern __shared__ __attribute__((aligned(16))) _Float16 smem16[];
    _Float16* hbuf = smem16;                              // [NS][L][SR][SH]   gathered h of the set's last phase
    _Float16* xin = hbuf + NS * L * SR * SH;              // [NS][2 parity][SR][SX]
    _Float16* own = xin + NS * 2 * SR * SX;               // [wave 4][L][SR][UPW]  fresh slice of this wave (wave-private)
    f32x4* bias_s = reinterpret_cast<f32x4*>(own + 4 * L * SR * UPW);   // [wave 4][L][NTW][lane 64]: the accumulators' start values
    int* ctl = reinterpret_cast<int*>(bias_s + 4 * L * NTW * 64);   // [0] abort, [1] class ticket, [2] last-out, [3] same XCD

    // control words (all zero between launches): [8 class tickets, one per 64-byte line][n_wg XCD words][done]
    unsigned* const class_ticket = p.xcc_slots + 64;
    unsigned* const xcc_words = p.xcc_slots + 64 + 8 * 16;
    const int cls = blockIdx.x & 7;
    unsigned my_xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(my_xcc));
    my_xcc &= 0xFu;
    // a launch that finds the sticky status word set (an earlier launch on this model aborted) leaves at once
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
    const int row0 = cluster * (NS * SR);
    if (tid == 0)
        __hip_atomic_store(xcc_words + cluster * GH + member, 0x10u | my_xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    // ---- weights: registers (binary16 pairs), for the whole launch; host layout [member16][wave][32-deep block][lane][8]
    //      with 16 units per "member16": this wave's two tiles are waves (2*wave, 2*wave + 1) of member16 = 2*member + (wave >> 1)
    f32x4 w0[NTW][NB0];                 // binary16 octets, held as 4-register tuples
    f32x4 w1[NTW][NB1];
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
        const int m16 = 2 * member + (wave >> 1), w16 = 2 * (wave & 1) + t;
        const f32x4* s0 = reinterpret_cast<const f32x4*>(p.wcl[0]) + ((size_t)(m16 * 4 + w16) * NB0) * 64 + lane;
#pragma unroll
        for (int i = 0; i < NB0; ++i) w0[t][i] = s0[i * 64];
        const f32x4* s1 = reinterpret_cast<const f32x4*>(p.wcl[1]) + ((size_t)(m16 * 4 + w16) * NB1) * 64 + lane;
#pragma unroll
        for (int i = 0; i < NB1; ++i) w1[t][i] = s1[i * 64];
    }
    // unit of (tile t, lane group g): member*32 + wave*8 + t*4 + g
    // (b_ih + b_hh, f32) of this lane's four gates per (layer, tile): the accumulators start from it; kept in LDS, not in
    // 16 registers -- the register file is the weight store
#pragma unroll
    for (int l = 0; l < L; ++l)
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            f32x4 bv;
#pragma unroll
            for (int k = 0; k < 4; ++k) bv[k] = p.bias[l][k * H + member * 32 + wave * 8 + t * 4 + g];
            bias_s[((wave * L + l) * NTW + t) * 64 + lane] = bv;
        }
    // per-set register state is kept as "this section's set" / "the other set" and swapped at the end of every section,
    // so the section body exists once (no unrolling over the sets, no dynamically indexed register arrays)
    float cst[L][NTW], cst_o[L][NTW];
#pragma unroll
    for (int l = 0; l < L; ++l)
#pragma unroll
        for (int t = 0; t < NTW; ++t) cst[l][t] = cst_o[l][t] = 0.0f;

    const __amdgpu_buffer_rsrc_t hx_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.hx, 0, (int)p.hx_bytes, 0x00020000);
    unsigned* const flags_of = p.xflags + (size_t)cluster * NS * NFL;      // [set][member*4 + wave] epoch = phases published
    constexpr unsigned SET_BYTES = PIECES * 16;                            // one (set, parity): [layer][member][wave][row][8 halves]
    auto hx_base = [&](int s, int par) -> unsigned { return (unsigned)((((size_t)cluster * NS + s) * 2 + par) * SET_BYTES); };

    // ---- x staging (f64 z-score, then binary16); thread owns NE elements of a set's [SR][KX] step slab ------------
    constexpr int NE = (SR * KX) / 256;
    const int xk = tid % KX;
    const double x_mean = (normalize && xk < I) ? p.xx_m[xk] : 0.0;
    const double x_std = (normalize && xk < I) ? p.xx_s[xk] : 1.0;
    const double x_rstd = (normalize && xk < I) ? p.xx_r[xk] : 1.0;     // 1 / std, rounded once on the host
    float xr[NE], xr_o[NE];
    // loads go through a buffer descriptor over this cluster's rows: rows past the batch and the padded columns k >= I
    // fall outside it and read as 0 (no predicates, no 64-bit per-lane addresses: one 32-bit offset per element)
    const int rows_here = bcast_x ? NS * SR : max(0, min(NS * SR, p.B - row0));
    // (the descriptor's words are forced into scalar registers: left to the compiler they end up in vector registers
    //  and every load becomes a readfirstlane waterfall loop)
    const unsigned long long x_addr = reinterpret_cast<unsigned long long>(p.x + (bcast_x ? (size_t)0 : (size_t)row0 * T * I));
    const unsigned x_lo = __builtin_amdgcn_readfirstlane((unsigned)x_addr), x_hi = __builtin_amdgcn_readfirstlane((unsigned)(x_addr >> 32));
    const int x_bytes = __builtin_amdgcn_readfirstlane((int)((size_t)(bcast_x ? 1 : rows_here) * T * I * sizeof(float)));
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<float*>(((unsigned long long)x_hi << 32) | x_lo), 0, x_bytes, 0x00020000);
    const unsigned x_rowbytes = bcast_x ? 0u : (unsigned)(T * I * sizeof(float));
    auto fetch_x = [&](float (&dst)[NE], int s, int t) {
        const int slot = (t + p.x_ring >= T) ? t + p.x_ring - T : t + p.x_ring;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int row = s * SR + (tid + 256 * e) / KX;
            const unsigned off = (xk < I && row < rows_here) ? (unsigned)row * x_rowbytes + (unsigned)(xk * sizeof(float)) : 0x80000000u;
            dst[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rsrc, off, (unsigned)(slot * I * sizeof(float)), 0));
        }
    };
    // (x - m) / s in f64, correctly rounded: q0 = d * (1/s), one residual step q0 + (d - q0 s)(1/s) -- the tail of the
    // hardware division sequence, bit-identical to the division (as in lstm_cluster.hip); (0, 1, 1) passes x through
    auto stage_x = [&](const float (&src)[NE], int s, int t) {
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int row = (tid + 256 * e) / KX;
            const double d = (double)src[e] - x_mean;
            const double q0 = d * x_rstd;
            const double rr = fma(-q0, x_std, d);
            const double q1 = fma(rr, x_rstd, q0);
            // f64 -> f32 -> binary16 in two roundings, as the reference path's float32 model input stored as binary16 (kept
            // apart by the empty asm: fused, the compiler emits a ~45-instruction software f64 -> f16 conversion)
            float xf = (float)((rr == rr) ? q1 : q0);
            asm volatile("" : "+v"(xf));
            xin[((s * 2 + (t & 1)) * SR + row) * SX + xk] = (_Float16)xf;
        }
    };
    fetch_x(xr, 0, 0);
    fetch_x(xr_o, 1, 0);
    stage_x(xr, 0, 0);
    stage_x(xr_o, 1, 0);
    if (T > 1) { fetch_x(xr, 0, 1); fetch_x(xr_o, 1, 1); }

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
    const bool in_l2 = ctl[3] != 0 && (p.flags & APE_DIAG_WRITE_THROUGH) == 0;     // uniform over the cluster

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
    // piece q = tid + 256 k of a set: byte q * 16 of the (set, parity) block = (layer, member, wave, row) in that order
    auto issue_gather = [&](int s, int par, f32x4 (&gv)[NGV]) {
        const unsigned base = hx_base(s, par);
#pragma unroll
        for (int k = 0; k < NGV; ++k)
            gv[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(hx_rsrc, base + (unsigned)((tid + 256 * k) * 16), 0, 16 /* sc1 */));
    };
    auto commit_gather = [&](int s, const f32x4 (&gv)[NGV]) {
#pragma unroll
        for (int k = 0; k < NGV; ++k) {
            const int q = tid + 256 * k;
            const int row = q & 15, wv = (q >> 4) & 3, mem = (q >> 6) & 7, l = q >> 9;
            *reinterpret_cast<f32x4*>(hbuf + ((s * L + l) * SR + row) * SH + mem * 32 + wv * 8) = gv[k];
        }
    };
    // the flag a wave owes for the slices it stored last (other set): raised once those stores have drained
    int pend_set = -1;
    unsigned pend_epoch = 0u;
    auto raise_pending = [&]() {              // caller has waited vmcnt(0)
        if (pend_set < 0) return;
        if (lane == 0) {
            unsigned* f = flags_of + pend_set * NFL + member * 4 + wave;
            if (in_l2) *reinterpret_cast<volatile unsigned*>(f) = pend_epoch;           // plain: stays in the XCD's L2
            else __hip_atomic_store(f, pend_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        pend_set = -1;
    };

    const int P = T + L - 1;
#ifdef APE_CLUSTER_STAMPS
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long st_begin = st_t0, st_rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    // Section (ph, s) = row set s in phase ph.  What a section needs from the exchange -- the slices its set published a
    // phase ago -- is PREFETCHED by the section in front of it (the other set's): that section looks at the flags
    // after its layer-1 MFMAs (one load per lane, in flight under the gate math) and, if every peer wave has
    // published, puts the whole gather into flight before its own publish store; the needing section then only waits for
    // loads that have had a section's tail to land, side by side with the drain of that publish store, whose flag goes up
    // right after the same wait.  If a peer was late the section falls back to the blocking form.
    bool prefetched = false;
    f32x4 gv[NGV];
#pragma unroll 1
    for (int sec = 0; sec < NS * (P + 1); ++sec) {     // phase P: only the final gather of both sets (for the head)
        {
            const int ph = sec >> 1, s = sec & 1;
            // ---- S0: bring in what this set's last phase published (nothing before phase 0) ----------------------------------
            if (ph > 0 && !prefetched && !d_noex) {               // first phases, a late peer, the final gathers
                wait_flags(s, (unsigned)ph);
                V2_STAMP(0);                                      // 0: blocking flag poll
                issue_gather(s, (ph - 1) & 1, gv);
            }
            // the gather (prefetched a section's tail ago, or just issued) and the other set's publish store, which in the
            // steady state was issued right behind the prefetch: both have been in flight side by side
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            V2_STAMP(1);                                          // 1: wait for the gather / the store drain
            if (!d_noex) raise_pending();                         // the flag owed for that store
            if (ph > 0 && !d_noex) commit_gather(s, gv);
            prefetched = false;
            V2_STAMP(2);                                          // 2: LDS commit
            __syncthreads();
            V2_STAMP(3);                                          // 3: barrier
            if (ctl[0] != 0) return;
            if (ph == P) continue;                                // gather-only tail (no per-set register state is needed any more)
            // this section's activation fragments (both layers read the LDS state of the set's last phase only): layer 0's
            // now, layer 1's once layer 0's registers are free -- they land under the gate math of layer 0
            f32x4 ax[QX], a0r[QH], a1i[QH], a1r[QH];
            if (ph < T) {
                load_frags<QX>(ax, xin + ((s * 2 + (ph & 1)) * SR + r) * SX + 8 * g);
                if (ph > 0) load_frags<QH>(a0r, hbuf + ((s * L + 0) * SR + r) * SH + 8 * g);
            }
            V2_STAMP(7);                                          // 7: fragment read issue
            // the next section: the other set, in this phase (s = 0) or the next (s = 1); it needs epoch `want`
            const int sn = s ^ 1, phn = ph + s;
            const unsigned want = (unsigned)phn;
            unsigned peek = want;

            // ---- every active layer of this set, back to back (layer l works on step t = ph - l) -------------------------
#pragma unroll
            for (int l = 0; l < L; ++l) {
                const int t = ph - l;
                const bool active = t >= 0 && t < T;              // uniform over the grid
                f32x4 acc[NTW];
                if (active) {
#pragma unroll
                    for (int tt = 0; tt < NTW; ++tt) acc[tt] = bias_s[((wave * L + l) * NTW + tt) * 64 + lane];
                    if (d_nomfma) {
                    } else if (l == 0) {
                        span<NTW, QX, NB0>(acc, ax, w0, 0);
                        if (t > 0) span<NTW, QH, NB0>(acc, a0r, w0, QX);
                    } else {
                        span<NTW, QH, NB1>(acc, a1i, w1, 0);
                        if (t > 0) span<NTW, QH, NB1>(acc, a1r, w1, QH);
                    }
                    mfma_drain<NTW>(acc);
                }
                V2_STAMP(4);                                      // 4: MFMA spans (incl. the wait for the LDS reads feeding them)
                if (l == 0) {
                    if (ph >= 1) load_frags<QH>(a1i, hbuf + ((s * L + 0) * SR + r) * SH + 8 * g);
                    if (ph > 1) load_frags<QH>(a1r, hbuf + ((s * L + 1) * SR + r) * SH + 8 * g);
                    // x of the next step: registers -> the other parity buffer (its readers are two sections back), next fetch
                    if (ph + 1 < T && !d_nox) {
                        stage_x(xr, s, ph + 1);
                        if (ph + 2 < T) fetch_x(xr, s, ph + 2);
                    }
                } else if (want > 0u && lane < NFL && !d_noex) {
                    // [B0] look at the flags the next section needs; the load flies under the gate math below
                    peek = __hip_atomic_load(flags_of + sn * NFL + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                V2_STAMP(2);
                if (!active) continue;
                // gates + cell update, lane-local: registers 0..3 = i,f,g,o of (unit tt*4 + g, batch row r)
#pragma unroll
                for (int tt = 0; tt < NTW; ++tt) {
                    float hval;
                    if (d_noact) {
                        const float c = acc[tt][1] * cst[l][tt] + acc[tt][0] * acc[tt][2];
                        cst[l][tt] = c;
                        hval = acc[tt][3] * c;
                    } else {
                        const float iv = gate_act(acc[tt][0], false), fv = gate_act(acc[tt][1], false);
                        const float gg = gate_act(acc[tt][2], true), ov = gate_act(acc[tt][3], false);
                        const float c = fv * cst[l][tt] + iv * gg;
                        cst[l][tt] = c;
                        hval = ov * gate_act(c, true);
                    }
                    own[((wave * L + l) * SR + r) * UPW + tt * 4 + g] = (_Float16)hval;
                }
                V2_STAMP(5);                                      // 5: gates + cell update + own-slice staging
            }
            // [B] every peer wave has published what the next section needs: its whole gather goes into flight now
            if (want > 0u && !d_noex && __all((int)(peek >= want))) {
                issue_gather(sn, (phn - 1) & 1, gv);
                prefetched = true;
            }
            // ---- publish: lanes 0..15 send layer 0's row, lanes 16..31 layer 1's (this wave's 8 units = 16 bytes each) ----
            //      (exactly ONE store instruction per wave: the counted wait at the top of the next section relies on it)
            {
                const int l = lane >> 4, row = lane & 15, t = ph - l;
                const bool live = lane < 16 * L && t >= 0 && t < T;
                const u32x4 hv = *reinterpret_cast<const u32x4*>(own + ((wave * L + (l & (L - 1))) * SR + row) * UPW);
                // dead lanes aim outside the buffer descriptor: the store instruction is issued by every wave, writes nothing there
                const unsigned off = (live && !d_noex) ? hx_base(s, ph & 1) + (unsigned)((((l * GH + member) * 4 + wave) * SR + row) * 16) : 0x80000000u;
                if (in_l2) __builtin_amdgcn_raw_buffer_store_b128(hv, hx_rsrc, off, 0, 0);
                else __builtin_amdgcn_raw_buffer_store_b128(hv, hx_rsrc, off, 0, 16 /* sc1: write-through */);
                pend_set = s;
                pend_epoch = (unsigned)(ph + 1);
            }
            V2_STAMP(6);                                          // 6: publish (LDS read + store issue)
            // the other set is next: swap the per-set register state
#pragma unroll
            for (int l = 0; l < L; ++l)
#pragma unroll
                for (int tt = 0; tt < NTW; ++tt) { const float tmp = cst[l][tt]; cst[l][tt] = cst_o[l][tt]; cst_o[l][tt] = tmp; }
#pragma unroll
            for (int e = 0; e < NE; ++e) { const float tmp = xr[e]; xr[e] = xr_o[e]; xr_o[e] = tmp; }
        }
    }
#ifdef APE_CLUSTER_STAMPS
    if (p.dbg_wg != nullptr && lane == 0 && wave == 0 && cluster == 0 && member == 0) {
        for (int k = 0; k < 8; ++k) p.dbg_wg[k] = st_acc[k];
        p.dbg_wg[8] = __builtin_amdgcn_s_memtime() - st_begin;
        p.dbg_wg[9] = __builtin_amdgcn_s_memrealtime() - st_rt0;
    }
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    raise_pending();

    // ---- head (f32 weights, f16 h): member m finishes rows 4m .. 4m+3 of the cluster's 32 ---------------------------
    {
        constexpr int RPM = (NS * SR) / GH;
        if (tid < RPM * O) {
            const int rr = tid / O, o = tid - rr * O;
            const int row = member * RPM + rr;                    // 0..31: set = row / 16
            const int b = row0 + row;
            if (b < p.B) {
                const _Float16* hv = hbuf + (((row >> 4) * L + (L - 1)) * SR + (row & 15)) * SH;
                const float* wv = p.w_out + (size_t)o * H;
                float sacc = 0.0f;
                for (int k = 0; k < H; ++k) sacc = fmaf((float)hv[k], wv[k], sacc);
                p.y[(size_t)b * O + o] = sacc + p.b_out[o];
            }
        }
    }
    // ---- self-cleaning: the last workgroup out re-zeroes every polled word ------------------------------------------
    __syncthreads();
    if (tid == 0)
        ctl[2] = (__hip_atomic_fetch_add(p.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) ? 1 : 0;
    __syncthreads();
    if (ctl[2] != 0) {
        const int n_flags = (int)(gridDim.x / GH) * NS * NFL;
        for (int i = tid; i < n_flags; i += 256) __hip_atomic_store(p.xflags + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int i = tid; i < (int)gridDim.x; i += 256) __hip_atomic_store(xcc_words + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid < 8) __hip_atomic_store(class_ticket + tid * 16, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid == 0) __hip_atomic_store(p.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <int H, int L, int KX>
constexpr size_t smem_bytes() {
    return ((size_t)2 * L * 16 * (H + 16) + (size_t)2 * 2 * 16 * (KX + 16) + (size_t)4 * L * 16 * 8) * sizeof(_Float16) +
           (size_t)4 * L * 2 * 64 * 16 + 16;
}

}  // namespace

bool ape_cluster_f16v2_supported(int H, int L, int KX) { return H == 256 && L == 2 && KX == 32; }

hipError_t ape_prepare_lstm_cluster_f16v2(int H, int L, int KX) {
    if (!ape_cluster_f16v2_supported(H, L, KX)) return hipSuccess;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&ape_lstm_cluster_f16v2<256, 2, 32>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, APE_LDS_BYTES);
}

// `clusters` = 32-row clusters needed; the grid is rounded up to whole block-index classes (8 clusters), the extra
// clusters own rows past the batch and only take part in the formation
hipError_t ape_launch_lstm_cluster_f16v2(int H, int L, int KX, int clusters, const ClusterParams& p, hipStream_t stream) {
    if (!ape_cluster_f16v2_supported(H, L, KX)) return hipErrorInvalidValue;
    const int grid_clusters = (clusters + 7) / 8 * 8;
    constexpr size_t smem = smem_bytes<256, 2, 32>();
    hipLaunchKernelGGL((ape_lstm_cluster_f16v2<256, 2, 32>), dim3(grid_clusters * 8), dim3(256), smem, stream, p);
    return hipGetLastError();
}
