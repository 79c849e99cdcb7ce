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
//   * the gathered slices never pass through registers: the exchange buffer's order [layer][member][wave][row][8 units]
//     IS the MFMA fragment order of the activation operand (k-block = member, k-group = wave), so a set's 16 KB are copied
//     global -> LDS by 16 LDS-DMA instructions per workgroup (`buffer_load_dwordx4 ... lds`, inline asm: outside the
//     compiler's s_waitcnt bookkeeping, waited for with a counted vmcnt), PREFETCHED by the section in front -- the other
//     set's -- which looks at the flags after its layer-1 MFMAs and starts the copy before its own publish store; the
//     needing section only waits for what is left of it, and nothing in the steady state waits for a store, a flag or a
//     gather right after issuing it.
//   * the steady-state section (both layers active, every condition constant) is compiled separately from the general
//     one (pipeline fill / drain, late peers): with one wave per SIMD every scalar branch instruction is on the critical path.
//   * self-cleaning like the other cluster kernels: the last workgroup out re-zeroes every polled word.
#include <type_traits>

#include "ape_internal.h"
#include "async_look.h"
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
    static_assert(NTW == 1 || NTW == 2, "one or two tiles per wave");
    if constexpr (NTW == 2) asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[1]));
    else asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[0]));
}

// NQ 32-deep k-blocks of LDS activations (16 rows x 32 k each) into registers; `stride` halves between k-blocks
template <int NQ>
__device__ __forceinline__ void load_frags(f32x4 (&a)[NQ], const _Float16* __restrict__ src, int stride) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) a[q] = *reinterpret_cast<const f32x4*>(src + stride * q);
}
// one LDS-DMA wave-instruction: 64 lanes x 16 bytes from the buffer (lane offset `voff`, uniform `soff`) to 1 KiB of LDS at
// the wave-uniform byte address `lds_addr`; sc1 = L1-bypassing, like every load of handed-off bytes.  M0 is written in the
// same statement that reads it (the compiler does not preserve it across statements).
__device__ __forceinline__ void dma_1k(unsigned lds_addr, unsigned voff, u32x4 rsrc, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen sc1 lds"
                 :: "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
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

// (the flag looks: LDS-DMA into the wave's landing zone, async_look.h -- no destination register)
// sweep positions of a section's hooks (tests/tools): where the flag owed for the other set's publish goes up (0: right behind the barrier,
// 1: behind layer 0's MFMAs) and where the look at the next section's flags is issued (0: behind layer 1's MFMAs, 1: behind layer 0's gate
// math, 2: behind layer 0's MFMAs, 3: between layer 1's two spans -- the default since round 5: the look now lands in LDS and is read back
// from there, one LDS hop more than the register destination it had; swept under the write-through hand-over at 1024 x 64:
// 0 241.3 us, 1 243.3, 2 288.0, 3 235.1; flag raise behind layer 0's MFMAs instead of behind the barrier 255.7)
#ifndef F16_LOOK_AT
#define F16_LOOK_AT 3
#endif

// UPW = 4 (round 4, "duo"): 16-unit members -- a wave owns 4 units = ONE tile, 100 weight registers -- so that a workgroup fits twice on a
// CU (<= 256 registers per wave, 47 KB of LDS): 16-member clusters of 32 rows, 512 workgroups for 1024 windows, TWO independent
// workgroups per CU with a barrier each.  A section of this kernel is ~48 % compute and ~52 % hand-over waits (profiles/r03_f16v2_stamps.md);
// a second, independent wave per SIMD can run under the first one's waits.  The pieces of the exchange are 8 bytes per wave then (two
// waves share a 16-byte k-group of the fragment order); everything else is the same code.
template <int H, int L, int KX, int UPW>
__global__ __launch_bounds__(256, (UPW == 4 ? 2 : 1)) void ape_lstm_cluster_f16v2(const ClusterParams p) {
    constexpr int NTW = UPW / 4;            // 16-column MFMA tiles per wave (column = unit * 4 + gate)
    constexpr int GH = H / (4 * UPW);       // members per cluster
    constexpr int SR = 16, NS = 2;          // rows per set, sets per cluster
    constexpr int SX = KX + 16;             // LDS row stride of the x slabs in halves (16-byte multiple, conflict-free b128 reads)
    constexpr int QX = KX / 32, QH = H / 32;
    constexpr int NB0 = QX + QH, NB1 = 2 * QH;
    constexpr int NFL = 4 * GH;             // flags per (cluster, set): one per member wave
    constexpr int KG = 4 * UPW / 8;         // 16-byte k-groups (8 units) per member: 4 (one per wave) or 2 (one per pair of waves)
    constexpr int PIECES = L * GH * KG * SR; // 16-byte pieces of one set's gathered slices
    constexpr int NDMA = PIECES / 256;      // LDS-DMA instructions per wave and gather
    constexpr int HL = GH * KG * SR * 8;    // halves of one (set, layer) block: [member][k-group][row][8 units]
    static_assert((UPW == 8 || UPW == 4) && GH * KG == 4 * QH && L == 2 && PIECES % 256 == 0 && NFL <= 64, "built for the 2 x 256 models");

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
    _Float16* hbuf = smem16;                              // [NS][L][member][k-group][row][8]: gathered h of the set's last phase, in
                                                          //  the exchange order = MFMA fragment order (k-block = member, k-group = wave)
    _Float16* xin = hbuf + NS * L * HL;                   // [NS][2 parity][SR][SX]
    _Float16* own = xin + NS * 2 * SR * SX;               // [wave 4][L][SR][UPW]  fresh slice of this wave (wave-private)
    f32x4* bias_s = reinterpret_cast<f32x4*>(own + 4 * L * SR * UPW);   // [wave 4][L][NTW][lane 64]: the accumulators' start values
    unsigned* look_s = reinterpret_cast<unsigned*>(bias_s + 4 * L * NTW * 64);   // [wave 4][64]: landing zones of the flag looks (async_look.h)
    int* ctl = reinterpret_cast<int*>(look_s + 4 * 64);             // [0] abort, [1] class ticket, [2] last-out, [3] same XCD
    float* xraw = reinterpret_cast<float*>(ctl + 4);                // [NS][2 parity][SR * KX] raw float32 step slabs (UPW == 8: fetched by LDS-DMA)

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
        const int m16 = (UPW == 8) ? 2 * member + (wave >> 1) : member, w16 = (UPW == 8) ? 2 * (wave & 1) + t : wave;
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
    // (round 6: ... and in registers too in the form with one wave per SIMD -- 328 of its 512 are in use --: the start values used to come out of
    //  LDS at the head of each layer's span, an exposed LDS latency in front of the first MFMA, twice a section)
    f32x4 bias_r[L][NTW];
#pragma unroll
    for (int l = 0; l < L; ++l)
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            f32x4 bv;
#pragma unroll
            for (int k = 0; k < 4; ++k) bv[k] = p.bias[l][k * H + member * (4 * UPW) + wave * UPW + t * 4 + g];
            bias_s[((wave * L + l) * NTW + t) * 64 + lane] = bv;
            bias_r[l][t] = bv;
        }
    // per-set register state is kept as "this section's set" / "the other set" and swapped at the end of every section,
    // so the section body exists once per kind (no unrolling over the sets, no dynamically indexed register arrays)
    float cst[L][NTW], cst_o[L][NTW];
#pragma unroll
    for (int l = 0; l < L; ++l)
#pragma unroll
        for (int t = 0; t < NTW; ++t) cst[l][t] = cst_o[l][t] = 0.0f;

    // exchange buffer: descriptor for the compiler's loads / stores, and the same words as a scalar tuple for the DMA asm
    const __amdgpu_buffer_rsrc_t hx_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.hx, 0, (int)p.hx_bytes, 0x00020000);
    const unsigned long long hx_addr = reinterpret_cast<unsigned long long>(p.hx);
    u32x4 hx_desc;
    hx_desc[0] = __builtin_amdgcn_readfirstlane((unsigned)hx_addr);
    hx_desc[1] = __builtin_amdgcn_readfirstlane((unsigned)(hx_addr >> 32) & 0xFFFFu);
    hx_desc[2] = __builtin_amdgcn_readfirstlane((unsigned)p.hx_bytes);
    hx_desc[3] = 0x00020000u;
    unsigned* const flags_of = p.xflags + (size_t)cluster * NS * NFL;      // [set][member*4 + wave] epoch = phases published
    const ape_desc_t fl_desc = ape_make_desc(p.xflags, (unsigned)((gridDim.x / GH) * NS * NFL * sizeof(unsigned)));
    const unsigned fl_off = (unsigned)(cluster * NS * NFL * sizeof(unsigned));
    const unsigned look_voff = (unsigned)((lane & (NFL - 1)) * sizeof(unsigned));
    const unsigned look_lds = (unsigned)reinterpret_cast<unsigned long long>(look_s) + (unsigned)(wave * 256);
    const unsigned* const look_mine = look_s + wave * 64 + lane;
    constexpr unsigned SET_BYTES = PIECES * 16;                            // one (set, parity): [layer][member][wave][row][8 halves]
    auto hx_base = [&](int s, int par) -> unsigned { return (unsigned)((((size_t)cluster * NS + s) * 2 + par) * SET_BYTES); };
    const unsigned hbuf_lds = (unsigned)reinterpret_cast<unsigned long long>(hbuf);     // LDS byte address (low half of the flat one)

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
    unsigned x_off[NE];                      // byte offset of this thread's elements inside set 0 (set 1: + SR rows)
#pragma unroll
    for (int e = 0; e < NE; ++e) x_off[e] = (unsigned)((tid + 256 * e) / KX) * x_rowbytes + (unsigned)(xk * sizeof(float));
    auto fetch_x = [&](float (&dst)[NE], int s, int t) {
        const int slot = (t + p.x_ring >= T) ? t + p.x_ring - T : t + p.x_ring;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int row = s * SR + (tid + 256 * e) / KX;
            const unsigned off = (xk < I && row < rows_here) ? x_off[e] + (unsigned)(s * SR) * x_rowbytes : 0x80000000u;
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
    // Round 6, the eight-unit form: a step slab travels global -> LDS as RAW float32 by LDS-DMA (two instructions per wave: element tid + 256 e
    // lands at its own index, fetched and later read by the same thread) and is z-scored from there at the TOP of the set's next section, in
    // front of the wait for the gathered slices -- ~0.15 us of float64 arithmetic per section that used to sit between the layers' MFMAs.
    // (With the slab in REGISTERS that did not work: hipcc's `s_waitcnt` in front of the first use of a compiler-issued load does not count
    //  the asm-issued look and gather behind it and waits for them too.  An LDS-DMA has no destination register and is waited for by a
    //  counted `vmcnt` written here.)
    const ape_desc_t x_desc = ape_make_desc(reinterpret_cast<const void*>(((unsigned long long)x_hi << 32) | x_lo), (unsigned)x_bytes);
    const unsigned xraw_lds = (unsigned)reinterpret_cast<unsigned long long>(xraw) + (unsigned)(wave * 256);
    auto dma_x = [&](int s, int t) {
        const int slot = (t + p.x_ring >= T) ? t + p.x_ring - T : t + p.x_ring;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int row = s * SR + (tid + 256 * e) / KX;
            const unsigned off = (xk < I && row < rows_here) ? x_off[e] + (unsigned)(s * SR) * x_rowbytes : 0x80000000u;      // (past the descriptor: 0)
            look_issue(xraw_lds + (unsigned)(((s * 2 + (t & 1)) * SR * KX + 256 * e) * sizeof(float)), off, x_desc, (unsigned)(slot * I * sizeof(float)));
        }
    };
    auto stage_x_lds = [&](int s, int t) {
        float src[NE];
#pragma unroll
        for (int e = 0; e < NE; ++e) src[e] = xraw[(s * 2 + (t & 1)) * SR * KX + tid + 256 * e];
        stage_x(src, s, t);
    };
    if constexpr (UPW == 8) {
        dma_x(0, 0);
        dma_x(1, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stage_x_lds(0, 0);
        stage_x_lds(1, 0);
        if (T > 1) { dma_x(0, 1); dma_x(1, 1); }
    } else {
        fetch_x(xr, 0, 0);
        fetch_x(xr_o, 1, 0);
        stage_x(xr, 0, 0);
        stage_x(xr_o, 1, 0);
        if (T > 1) { fetch_x(xr, 0, 1); fetch_x(xr_o, 1, 1); }
    }

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
        // hand-over form (ape_internal.h): write-through (`sc1`) payload stores unless the caller opted into the plain in-XCD form AND the
    // members were verified to share an XCD; uniform over the cluster
    const bool in_l2 = APE_HANDOVER_IN_L2(p.flags, ctl[3] != 0);

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
    // a set's 16 KB straight from the exchange buffer into its LDS block: wave w copies KiB w, w + 4, w + 8, w + 12
    const unsigned dma_voff = (unsigned)(lane * 16);
    auto issue_gather = [&](int s, int par) {
        const unsigned src = hx_base(s, par) + (unsigned)(wave * 1024), dst = hbuf_lds + (unsigned)(s * L * HL * 2 + wave * 1024);
#pragma unroll
        for (int k = 0; k < NDMA; ++k) dma_1k(dst + (unsigned)(k * 4096), dma_voff, hx_desc, src + (unsigned)(k * 4096));
    };
    // the flag a wave owes for the slices it stored last: raised once that store has drained
    int pend_set = -1;
    unsigned pend_epoch = 0u;
    auto raise_pending = [&]() {              // caller has waited vmcnt(0)
        if (pend_set < 0) return;
        if (lane == 0) {
            unsigned* f = flags_of + pend_set * NFL + member * 4 + wave;
            // always the agent-scope (sc1) store, also when the payload stays in the XCD's L2: a plain flag store takes
            // its time to leave the CU, and the peers look at these words early to prefetch (measured: 252 vs 289 us)
            __hip_atomic_store(f, pend_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // (pend_set stays: the waves of a member may take different paths to the barrier, and behind it wave 0 raises for all of them)
    };
    // the same behind the section's barrier, in front of which EVERY wave of the member has drained its store: wave 0 raises the member's
    // four words in one store instruction (round 6: the 32 per-wave flag stores of a cluster to one cache line complete one after the other
    // on the memory side -- lstm_cluster16.hip, profiles/r06_flag_serialisation.md; a quarter of the stores here)
    auto raise_member = [&]() {
        if (pend_set < 0) return;
        if (wave == 0 && lane < 4)
            __hip_atomic_store(flags_of + pend_set * NFL + member * 4 + lane, pend_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pend_set = -1;
    };
    // workgroup barrier that waits for this wave's LDS traffic only (not for the publish store or a DMA in flight)
    auto bar = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    const int P = T + L - 1;
#ifdef APE_CLUSTER_STAMPS
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long st_begin = st_t0, st_rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    // Section (ph, s) = row set s in phase ph; the sets alternate.  Per wave the vector-memory queue of a steady-state
    // section is, in issue order:  [publish store of the section in front]  x fetch (2 loads)  flag look (1 load)
    // gather DMA of the NEXT section's set (4)  publish store (1) -- so at the top of a section everything but the
    // youngest entry, the publish store, is waited for (`vmcnt(1)`: the set's slices are in LDS); behind the barrier the
    // store itself is waited for and its flag goes up.
    bool prefetched = false;
    auto section = [&](auto steady_tag, const int ph, const int s) -> bool {
        constexpr bool ST = decltype(steady_tag)::value;          // steady state: 2 <= ph <= T - 3, every condition below holds
        // x of this set's next step: raw slab (LDS, fetched by this set's section in front) -> z-score -> the other parity buffer of xin (its
        // readers are two sections back).  In FRONT of the wait for the gather.  Behind the slab's two DMAs (issued by this set's section in
        // front, i.e. the section before last) every section issues at least a look and a publish store, and the section in front its own two
        // slab DMAs first: at least SIX younger operations, whatever was prefetched -- so with six left in flight the slab has landed by
        // construction, while the youngest six of the usual case (look, four gather DMAs, publish store) may still be travelling.
        // (From REGISTERS this was measured and lost, 238.8 against 236.1 us: hipcc's own wait in front of the first use of a compiler-issued
        //  load does not count the asm-issued look and gather behind it, so it waited for them too.)
        if constexpr (UPW == 8) {
            if ((ST || ph + 1 < T) && !d_nox) {
                if (ST) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                stage_x_lds(s, ph + 1);
            }
        }
        // ---- S0: this set's slices of the last phase into LDS ------------------------------------------------------------
        if (ST || ph > 0) {
            if (!prefetched && !d_noex) {                         // first phases, a late peer, the final gathers
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                raise_pending();
                wait_flags(s, (unsigned)ph);
                V2_STAMP(0);                                      // 0: blocking flag poll
                issue_gather(s, (ph - 1) & 1);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the prefetched copy AND this wave's publish store (raise_member below)
            }
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        prefetched = false;
        V2_STAMP(1);                                              // 1: wait for the gather
        bar();
        V2_STAMP(3);                                              // 3: barrier
        if (!ST && ph == P) {                                     // gather-only tail (no per-set register state is needed any more)
            raise_member();
            return ctl[0] == 0;
        }
        // the flag owed for the OTHER set's publish store, issued at the end of the section in front: right behind the
        // barrier, so that the peers' look at these words (after THEIR layer-1 MFMAs) finds it.  Measured on configs[4]:
        // here 244 us; after the layer-0 MFMAs (the store certainly drained, nothing waits) 252 us; after layer 0's gate
        // math 292 us -- the later the flag, the more gathers miss their prefetch and fall back to the blocking form
        // (the sweep switch for the second position is gone: round 6's counted wait in front of the slab staging assumes THIS order of the
        //  wave's vector-memory operations -- with the raise behind layer 0's MFMAs the re-sweep read 228 us and wrong values)
        if (!d_noex) raise_member();
        const int abort_word = ctl[0];                            // read with the fragments, looked at before the publish
        // this section's activation fragments (both layers read the LDS state of the set's last phase only): layer 0's
        // now, layer 1's once layer 0's registers are free -- they land under the gate math of layer 0
        // (round 6: h^0 of the set's last phase is BOTH layer 0's recurrent operand and layer 1's input -- read once, kept through layer 0's
        //  gate math: the second read of the same 8 fragments cost every wave 8 of its 25 ds_read_b128 per section, 32 KB of LDS traffic per CU)
        f32x4 ax[QX], a0r[QH], a1r[QH];
        const _Float16* hset = hbuf + s * L * HL + g * 128 + r * 8;
        if (ST || ph < T) load_frags<QX>(ax, xin + ((s * 2 + (ph & 1)) * SR + r) * SX + 8 * g, 32);
        if (ST || ph > 0) load_frags<QH>(a0r, hset, 512);
        V2_STAMP(7);                                              // 7: fragment read issue
        // the next section: the other set, in this phase (s = 0) or the next (s = 1); it needs epoch `want`
        const int sn = s ^ 1, phn = ph + s;
        const unsigned want = (unsigned)phn;
        unsigned peek = want;
        bool peeked = false;

        // ---- every active layer of this set, back to back (layer l works on step t = ph - l) -----------------------------
#pragma unroll
        for (int l = 0; l < L; ++l) {
            const int t = ph - l;
            const bool active = ST || (t >= 0 && t < T);          // uniform over the grid
            f32x4 acc[NTW];
            if (active) {
#pragma unroll
                for (int tt = 0; tt < NTW; ++tt) acc[tt] = (UPW == 8) ? bias_r[l][tt] : bias_s[((wave * L + l) * NTW + tt) * 64 + lane];
                if (d_nomfma) {
                } else if (l == 0) {
                    span<NTW, QX, NB0>(acc, ax, w0, 0);
                    if (ST || t > 0) span<NTW, QH, NB0>(acc, a0r, w0, QX);
                } else {
                    span<NTW, QH, NB1>(acc, a0r, w1, 0);
                    if (F16_LOOK_AT == 3) {                           // (sweep position: between layer 1's two spans)
                        look_issue(look_lds, look_voff, fl_desc, fl_off + (unsigned)(sn * NFL * sizeof(unsigned)));
                        peeked = true;
                    }
                    if (ST || t > 0) span<NTW, QH, NB1>(acc, a1r, w1, QH);
                }
                mfma_drain<NTW>(acc);
            }
            V2_STAMP(4);                                          // 4: MFMA spans (incl. the wait for the LDS reads feeding them)
            if (l == 0 && F16_LOOK_AT == 2) {
                look_issue(look_lds, look_voff, fl_desc, fl_off + (unsigned)(sn * NFL * sizeof(unsigned)));
                peeked = true;
            }
            if (l == 0) {
                if (ST || ph > 1) load_frags<QH>(a1r, hset + HL, 512);
                // x of the next step: registers -> the other parity buffer (its readers are two sections back), next fetch
                if ((ST || ph + 1 < T) && !d_nox) {
                    if constexpr (UPW == 8) {
                        if (ST || ph + 2 < T) dma_x(s, ph + 2);
                    } else {
                        stage_x(xr, s, ph + 1);
                        if (ST || ph + 2 < T) fetch_x(xr, s, ph + 2);
                    }
                }
            } else if (!peeked) {
                // [B0] look at the flags the next section needs (if the section has not issued it yet: F16_LOOK_AT, or a layer-1 part that did not run); the load flies under the gate math below.  Issued as LDS-DMA into the wave's
                // landing zone (async_look.h) and read back at [B]: hipcc hoists the comparison of a compiler-visible load up to the load and
                // waits `vmcnt(0)` right behind it -- the L2 round trip it was meant to hide (round 3, found in the disassembly).
                look_issue(look_lds, look_voff, fl_desc, fl_off + (unsigned)(sn * NFL * sizeof(unsigned)));
                peeked = true;
            }
            V2_STAMP(2);
            // gates + cell update, lane-local: registers 0..3 = i,f,g,o of (unit tt*4 + g, batch row r)
#pragma unroll
            for (int tt = 0; active && tt < NTW; ++tt) {
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
            V2_STAMP(5);                                          // 5: gates + cell update + own-slice staging
            if (l == 0 && F16_LOOK_AT == 1) {                     // (sweep position: behind layer 0's gate math)
                look_issue(look_lds, look_voff, fl_desc, fl_off + (unsigned)(sn * NFL * sizeof(unsigned)));
                peeked = true;
            }
        }
        if (abort_word != 0) return false;                        // (a wave of this workgroup gave up in a blocking wait)
        // [B] every peer wave has published what the next section needs: its whole gather goes into flight now
        // (the landed look and this wave's fresh slices are read out of LDS together: one LDS latency in front of the judge, not two)
        u32x4 pub_v = {0u, 0u, 0u, 0u};
        if (peeked) look_landed();
        if constexpr (UPW == 8) pub_v = *reinterpret_cast<const u32x4*>(own + ((wave * L + ((lane >> 4) & (L - 1))) * SR + (lane & 15)) * UPW);
        if (peeked) peek = *look_mine;
        if ((ST || want > 0u) && !d_noex && __all((int)(peek >= want))) {
            issue_gather(sn, (phn - 1) & 1);
            prefetched = true;
        }
        // ---- publish: lanes 0..15 send layer 0's row, lanes 16..31 layer 1's (this wave's 8 units = 16 bytes each) ----
        //      (exactly ONE store instruction per wave: the counted wait at the top of the next section relies on it)
        {
            const int l = lane >> 4, row = lane & 15, t = ph - l;
            const bool live = lane < 16 * L && (ST || (t >= 0 && t < T));
            // dead lanes aim outside the buffer descriptor: the store instruction is issued by every wave, writes nothing there
            if constexpr (UPW == 8) {
                const u32x4 hv = pub_v;
                const unsigned off = (live && !d_noex) ? hx_base(s, ph & 1) + (unsigned)((((l * GH + member) * 4 + wave) * SR + row) * 16) : 0x80000000u;
                if (in_l2) __builtin_amdgcn_raw_buffer_store_b128(hv, hx_rsrc, off, 0, 0);
                else __builtin_amdgcn_raw_buffer_store_b128(hv, hx_rsrc, off, 0, 16 /* sc1: write-through */);
            } else {
                // this wave's 4 units are half of a 16-byte k-group: 8 bytes at (wave & 1) * 8 inside the piece of (member, wave >> 1, row)
                typedef unsigned u32x2 __attribute__((__vector_size__(2 * sizeof(unsigned))));
                const u32x2 hv = *reinterpret_cast<const u32x2*>(own + ((wave * L + (l & (L - 1))) * SR + row) * UPW);
                const unsigned off = (live && !d_noex)
                    ? hx_base(s, ph & 1) + (unsigned)((((l * GH + member) * KG + (wave >> 1)) * SR + row) * 16 + (wave & 1) * 8) : 0x80000000u;
                if (in_l2) __builtin_amdgcn_raw_buffer_store_b64(hv, hx_rsrc, off, 0, 0);
                else __builtin_amdgcn_raw_buffer_store_b64(hv, hx_rsrc, off, 0, 16 /* sc1: write-through */);
            }
            pend_set = s;
            pend_epoch = (unsigned)(ph + 1);
        }
        V2_STAMP(6);                                              // 6: publish (LDS read + store issue)
        // the other set is next: swap the per-set register state
#pragma unroll
        for (int l = 0; l < L; ++l)
#pragma unroll
            for (int tt = 0; tt < NTW; ++tt) { const float tmp = cst[l][tt]; cst[l][tt] = cst_o[l][tt]; cst_o[l][tt] = tmp; }
#pragma unroll
        for (int e = 0; e < NE; ++e) { const float tmp = xr[e]; xr[e] = xr_o[e]; xr_o[e] = tmp; }
        return true;
    };
#pragma unroll 1
    for (int sec = 0; sec < NS * (P + 1); ++sec) {     // phase P: only the final gather of both sets (for the head)
        const int ph = sec >> 1, s = sec & 1;
        const bool ok = (ph >= 2 && ph <= T - 3) ? section(std::true_type{}, ph, s) : section(std::false_type{}, ph, s);
        if (!ok) return;
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
                const _Float16* hv = hbuf + ((row >> 4) * L + (L - 1)) * HL + (row & 15) * 8;    // unit k at [k / 8][row][k % 8]
                const float* wv = p.w_out + (size_t)o * H;
                float sacc = 0.0f;
                for (int k = 0; k < H; ++k) sacc = fmaf((float)hv[(k >> 3) * 128 + (k & 7)], wv[k], sacc);
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

template <int H, int L, int KX, int UPW>
constexpr size_t smem_bytes() {
    return ((size_t)2 * L * 16 * H + (size_t)2 * 2 * 16 * (KX + 16) + (size_t)4 * L * 16 * UPW) * sizeof(_Float16) +
           (size_t)4 * L * (UPW / 4) * 64 * 16 + (size_t)4 * 64 * sizeof(unsigned) + 16 + (UPW == 8 ? (size_t)2 * 2 * 16 * KX * sizeof(float) : 0);
}

}  // namespace

bool ape_cluster_f16v2_supported(int H, int L, int KX) { return H == 256 && L == 2 && KX == 32; }

hipError_t ape_prepare_lstm_cluster_f16v2(int H, int L, int KX) {
    if (!ape_cluster_f16v2_supported(H, L, KX)) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ape_lstm_cluster_f16v2<256, 2, 32, 8>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, APE_LDS_BYTES);
    if (e != hipSuccess) return e;
    static_assert(2 * smem_bytes<256, 2, 32, 4>() <= APE_LDS_BYTES, "two workgroups of the 16-unit-member form per CU");
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&ape_lstm_cluster_f16v2<256, 2, 32, 4>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_bytes<256, 2, 32, 4>());
}

// `clusters` = 32-row clusters needed; the grid is rounded up to whole block-index classes (8 clusters), the extra
// clusters own rows past the batch and only take part in the formation.  `duo`: the 16-unit-member form, 16 workgroups per cluster,
// two per CU.
hipError_t ape_launch_lstm_cluster_f16v2(int H, int L, int KX, int clusters, const ClusterParams& p, hipStream_t stream, bool duo) {
    if (!ape_cluster_f16v2_supported(H, L, KX)) return hipErrorInvalidValue;
    const int grid_clusters = (clusters + 7) / 8 * 8;
    if (duo) {
        constexpr size_t smem = smem_bytes<256, 2, 32, 4>();
        hipLaunchKernelGGL((ape_lstm_cluster_f16v2<256, 2, 32, 4>), dim3(grid_clusters * 16), dim3(256), smem, stream, p);
    } else {
        constexpr size_t smem = smem_bytes<256, 2, 32, 8>();
        hipLaunchKernelGGL((ape_lstm_cluster_f16v2<256, 2, 32, 8>), dim3(grid_clusters * 8), dim3(256), smem, stream, p);
    }
    return hipGetLastError();
}
