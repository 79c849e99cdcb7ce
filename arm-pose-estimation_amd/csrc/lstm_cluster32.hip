// Weight-stationary cluster LSTM kernel, second generation, exact float32 (BASELINE.json configs[2]/[3]: 1024 windows
// x 64 frames per GPU).  Same arithmetic as lstm_cluster.hip / lstm_tile16.hip up to float32 summation order
// (reference estimate/nn_models.py:169-174,180-189: L stacked LSTM layers, gates i,f,g,o, zero initial state, Linear
// head on the last step), restructured around what the first-generation kernel loses (~9 % in the
// instructions between its 16x16x4 MFMAs, ~9 % in exchange work and waits, ~5 % in the gate math):
//
//   * v_mfma_f32_32x32x2_f32 on ONE accumulator tile per wave: 32 gate columns (8 hidden units x i,f,g,o) x 32 windows.
//     Same 64 FLOP/cycle/SIMD as the 16x16x4 form, but half the instructions and half the activation reads per
//     FLOP, and the MFMAs of a layer-step form one dependent chain (64-cycle issue = 64-cycle dependent latency), so the
//     one ds_read_b128 + s_waitcnt per four MFMAs, the scalar bookkeeping and the exchange instructions issue inside the
//     shadow of the running MFMA instead of between two of them.
//   * 8 members per cluster instead of 16: a member owns 32 hidden units of both layers, a wave 8 of them; its slice of
//     [W_ih | W_hh] is 400 registers per lane -- layer 1's 256 in the accumulator file, layer 0's 144 in architectural
//     VGPRs -- resident for the whole launch.  A cluster owns 32 windows, 32 clusters fill the 256 CUs.  Half the peers,
//     half the flags, 32 KB instead of 64 KB gathered per layer-step and member.
//   * column order inside a wave's tile is gate * 8 + unit: with the weights as the A operand, lane (window n = lane & 31,
//     half hh = lane >> 5) then holds the four gates of units 4 hh .. 4 hh + 3 of window n in its sixteen accumulator
//     registers -- the cell update is lane-local, and the four fresh h values ARE one 16-byte piece of the exchange layout
//     [member][wave][window][8 units]: the publish is one store from registers, no staging.
//   * that exchange layout is also the MFMA fragment order of the activation operand (k-block of 8 = one wave's units, a
//     lane reads its window's half row), so gathered slices never pass through registers: a layer-step's 32 KB are copied
//     global -> LDS by 32 LDS-DMA instructions per workgroup (`buffer_load_dwordx4 ... lds`, inline asm, outside the
//     compiler's s_waitcnt bookkeeping), PREFETCHED under the other layer's section: layers are software-pipelined
//     (phase p: layer l works on step p - l), every section reads only what was published at least two sections ago, and
//     the section in front looks at the flags half-way through its MFMAs and starts the copy.
//   * XCD-pure clusters where the dispatcher allows it (membership by arrival ticket within the block-index class
//     blockIdx % 8, the XCD each member really runs on verified at run time): payload by plain stores that stay in the
//     XCD's L2, else sc1 write-through stores; flags always agent-scope stores; loads of handed-off bytes always sc1
//     (MI355X guide G16 / visibility table row 1).  Bounded spins, sticky status word, self-cleaning -- as
//     lstm_cluster_f16v2.hip, whose exchange this is.
// Eval mode, last-step output (what the estimators' batched path and the benchmark use); dropout, all-steps output and
// other shapes stay on the first-generation kernels.
#include <type_traits>

#include "ape_internal.h"
#include "async_look.h"
#include "../../include/ape_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((__vector_size__(4 * sizeof(unsigned))));
constexpr unsigned SPIN_LIMIT = 1u << 22;

// diagnostic builds: timeline of a short launch (cluster 0, member 0, thread 0) in dbg_wg[256 + i] -- 0 entry, 1 rendezvous done, 2 step 0
// of layer 0 and the weights done, 8 + 2 * (2 * ph + l) / + 1: section (ph, l) behind its top barrier / at its end, 4 final gather, 5 head
#ifdef APE_CLUSTER_STAMPS
#define C32_TL(i) do { if (p.dbg_wg != nullptr && tid == 0 && cluster == 0 && member == 0 && (i) < 200) p.dbg_wg[256 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define C32_TL(i) do {} while (0)
#endif

// hardware v_exp_f32 (2^x) and v_rcp_f32, both ~1 ulp -- the formulas of the other kernels (lstm_cluster_common.h)
__device__ __forceinline__ float sigm(float v) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v)); }
__device__ __forceinline__ float tanh_(float v) { return 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.885390081777927f * v)) - 1.0f; }

// v_mfma_f32_32x32x2_f32 with the weight operand (A) in the accumulator file (AG) or in an architectural VGPR.
// hipcc does not model an asm MFMA's result hazard: the accumulators are read only behind mfma_drain(), which takes
// them as in/out operands so that nothing that reads them can be scheduled above it.
template <bool AG>
__device__ __forceinline__ void mfma32(f32x16& acc, float w, float a) {
    if constexpr (AG) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "a"(w), "v"(a));
    else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "v"(a));
}
__device__ __forceinline__ void mfma_drain(f32x16& acc) { asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc)); }

// NB k-blocks of 8: acc += W (registers w[w0 ...]) x activations (LDS, one ds_read_b128 per block: this lane's window,
// units 4 hh .. 4 hh + 3 of the block; `stride` floats between blocks), fragments fetched two blocks ahead.
// `mid(kb)` runs after the MFMAs of block kb (kb a constant after unrolling): the caller hangs exchange work there.
template <int NB, bool AG, int NW, typename Mid>
__device__ __forceinline__ void span32(f32x16& acc, const float* __restrict__ src, int stride, const float (&w)[NW], int w0, Mid&& mid) {
    f32x4 a0 = *reinterpret_cast<const f32x4*>(src);
    f32x4 a1 = (NB > 1) ? *reinterpret_cast<const f32x4*>(src + stride) : a0;
    f32x4 a2 = a1;
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
        if (kb + 2 < NB) a2 = *reinterpret_cast<const f32x4*>(src + stride * (kb + 2));
#pragma unroll
        for (int j = 0; j < 4; ++j) mfma32<AG>(acc, w[w0 + 4 * kb + j], a0[j]);
        mid(kb);
        a0 = a1;
        a1 = a2;
    }
}

// one LDS-DMA wave-instruction: 64 lanes x 16 bytes from the buffer (lane offset `voff`, uniform `soff`) to 1 KiB of LDS at
// the wave-uniform byte address `lds_addr`; sc1 = L1-bypassing, like every load of handed-off bytes.  M0 is written in the
// same statement that reads it (the compiler does not preserve it across statements).
// HOT = inside a section's MFMA stream, where every scalar instruction between two MFMAs is paid for (round 4, 1024 x 64: `s_nop 3`
// instead of `s_nop 0` in the eight copies of a section +4.7 us per launch, one more s_add per copy +5.8 us): there `soff` comes straight
// from scalar arithmetic.  The copies of the blocking forms (pipeline fill, final gather) are not in anybody's way, and there hipcc does
// reload `soff` from a spill lane (v_readlane_b32) right in front of the statement -- a VALU write of an SGPR needs five wait states
// before a vector-memory instruction reads it, and hipcc does not look inside an asm statement.  tools/check_mfma_hazards.py scans
// the build for exactly that adjacency, so a reload that turns up in front of a HOT copy fails the build.
// The HOT form therefore takes the two addresses WITHOUT the piece's offset and adds it as an immediate inside the statement: no
// per-piece sums for hipcc to hoist out of the phase loop (with three more section forms in round 4 they no longer fitted the scalar
// registers), two SALU instructions per copy, and the second s_add is the wait state the M0 write needs in front of the LDS-DMA.
template <bool HOT>
__device__ __forceinline__ void dma_1k(unsigned lds_addr, unsigned voff, u32x4 rsrc, unsigned soff, int piece_bytes) {
    if constexpr (HOT) {
        unsigned tmp;
        asm volatile("s_add_i32 m0, %1, %5\n\ts_add_i32 %0, %4, %5\n\tbuffer_load_dwordx4 %2, %3, %0 offen sc1 lds"
                     : "=&s"(tmp) : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff), "i"(piece_bytes) : "memory");
    } else {
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 3\n\tbuffer_load_dwordx4 %1, %2, %3 offen sc1 lds"
                     :: "s"(lds_addr + (unsigned)piece_bytes), "v"(voff), "s"(rsrc), "s"(soff + (unsigned)piece_bytes) : "memory");
    }
}

// A flag look that does NOT stall the MFMA stream: hipcc hoists the comparison of a compiler-visible load up to the load and puts
// `s_waitcnt vmcnt(0)` right behind it -- an L2 round trip exposed in every section (found in round 3 in the disassembly of this
// kernel as round 2 shipped it).  The look is issued as LDS-DMA into the wave's landing zone (async_look.h: no destination register;
// rounds 3-4 had one, and round 5 found what hipcc may do to it), read back from LDS one k-block in front of the judge.

// ENDS: the forms for the ends of a launch (step 0 of layer 0 in front of the bulk of the weights, MODE 1 / MODE 2 sections, see there) --
// the instantiation for SHORT windows (T <= APE_C32_ENDS_MAX_T; every deployed model has T = 6 or 8).  The timeline stamps
// (tests/tools/timeline_c32.py) show what they do at T = 6: layer 0's second step starts 1.5 us behind the weights instead of a whole
// exchange behind them, layer 0's third step and layer 1's last one find their slices in LDS (top waits 0.2-0.3 us).  Same-box A/B,
// 1024 x 6: 83.8 -> 82.3 us (best blocks; 87.2 -> 85.6 medians).  But with these forms in the kernel hipcc's code for the steady-state
// sections comes out ~0.3-1 % slower per phase (register allocation / layout: 12.24 vs 12.27 us per phase; 1024 x 32: 402 -> 406 us; 1024 x 12: no difference left), so
// longer windows keep the instantiation without them.
// hook positions of a steady-state section (k-block indices; see `mid` in the kernel): QF = where the flag owed for the section in front's publish
// store goes up, QPD = shift of the look / judge / gather positions from the middle of the section
#ifndef C32_QF
#define C32_QF 3
#endif
#ifndef C32_QPD
#define C32_QPD 0
#endif
template <int H, int L, int KX, bool ENDS>
__global__ __launch_bounds__(256, 1) void ape_lstm_cluster32(const ClusterParams p) {
    constexpr int UPW = 8;                  // hidden units per wave (x 4 gates = the 32 columns of its tile)
    constexpr int GH = H / (4 * UPW);       // members per cluster
    constexpr int MR = 32;                  // windows per cluster (one 32-row tile)
    constexpr int SX = KX + 4;              // LDS row stride of the x slab (floats): conflict-free ds_read_b128 (rows 36 dwords apart)
    constexpr int BX = KX / 8, BH = H / 8;  // k-blocks of 8
    constexpr int NW0 = 4 * (BX + BH);      // weight registers per lane, layer 0 (architectural VGPRs)
    constexpr int NW1 = 4 * (2 * BH);       //                            layer 1 (accumulator file)
    constexpr int NFL = 4 * GH;             // flags per (cluster, layer): one per member wave
    constexpr int HL = GH * 4 * MR * 8;     // floats of one gathered slice set [member][wave][window][8 units] = 32 KB
    constexpr int NDMA = HL * 4 / 1024 / 4; // LDS-DMA instructions per wave and gather
    static_assert(H == 256 && L == 2 && KX == 32 && GH == 8 && NDMA == 8, "built for the 2 x 256 models");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, hh = lane >> 5;     // window of the tile, half (units 4 hh .. 4 hh + 3 of the wave's 8)
    const int T = p.T, I = p.I, O = p.O;
    const bool normalize = (p.flags & APE_FLAG_NORMALIZE_INPUT) != 0;
    const bool bcast_x = (p.flags & APE_FLAG_BROADCAST_X) != 0;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* hb0 = smem;                      // [2 parity][HL]  h of layer 0, step parity (read by layer 0's recurrence AND layer 1's input)
    float* hb1 = hb0 + 2 * HL;              // [HL]           h of layer 1
    float* xin = hb1 + HL;                  // [MR][SX]
    f32x4* bias_s = reinterpret_cast<f32x4*>(xin + MR * SX);     // [wave 4][L][4 gates][lane 64]: the accumulators' start values
    unsigned* look_s = reinterpret_cast<unsigned*>(bias_s + 4 * L * 4 * 64);   // [wave 4][64]: landing zones of the flag looks (async_look.h)
    int* ctl = reinterpret_cast<int*>(look_s + 4 * 64);          // [0] abort, [1] class ticket, [2] last-out, [3] same XCD

    // control words (all zero between launches): [8 class tickets, one per 64-byte line][n_wg XCD words]
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
    const int row0 = cluster * MR;
    C32_TL(0);
    if (tid == 0)
        __hip_atomic_store(xcc_words + cluster * GH + member, 0x10u | my_xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    // ---- x: the first step's loads go out in front of the weights (loads return in order) -----------------------------
    constexpr int NE = (MR * KX) / 256;     // (window, k) elements of the step slab per thread, all with the same k
    const int xk = tid % KX, xrow = tid / KX;
    const int rows_here = bcast_x ? MR : max(0, min(MR, p.B - row0));
    // (descriptor words forced into scalar registers: in vector registers every load becomes a readfirstlane waterfall loop)
    const unsigned long long x_addr = reinterpret_cast<unsigned long long>(p.x + (bcast_x ? (size_t)0 : (size_t)row0 * T * I));
    const unsigned x_lo = __builtin_amdgcn_readfirstlane((unsigned)x_addr), x_hi = __builtin_amdgcn_readfirstlane((unsigned)(x_addr >> 32));
    const int x_bytes = __builtin_amdgcn_readfirstlane((int)((size_t)(bcast_x ? 1 : rows_here) * T * I * sizeof(float)));
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<float*>(((unsigned long long)x_hi << 32) | x_lo), 0, x_bytes, 0x00020000);
    const unsigned x_rowbytes = bcast_x ? 0u : (unsigned)(T * I * sizeof(float));
    // rows past the batch and the padded columns k >= I lie outside the descriptor and read as 0 (no predicates)
    const unsigned x_off0 = (xk < I) ? (unsigned)xrow * x_rowbytes + (unsigned)(xk * sizeof(float)) : 0x80000000u;
    const unsigned x_estride = (unsigned)(256 / KX) * x_rowbytes;
    float xr[NE];
    auto fetch_x = [&](int t) {
        const int slot = (t + p.x_ring >= T) ? t + p.x_ring - T : t + p.x_ring;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const unsigned off = (xrow + e * (256 / KX) < rows_here) ? x_off0 + (unsigned)e * x_estride : 0x80000000u;
            xr[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rsrc, off, (unsigned)(slot * I * sizeof(float)), 0));
        }
    };
    // f64 z-score, cast f32 (estimator.py:103-104, watch_phone_pocket_nn.py:100): (x - m) / s correctly rounded via the
    // host-rounded reciprocal and one residual step (bit-identical to the division, as in lstm_cluster.hip)
    double x_mean = (normalize && xk < I) ? p.xx_m[xk] : 0.0;
    double x_std = (normalize && xk < I) ? p.xx_s[xk] : 1.0;
    double x_rstd = (normalize && xk < I) ? p.xx_r[xk] : 1.0;
    // (opaque from here on: with more section forms in the kernel hipcc preferred to RE-LOAD the three constants inside the staging hook
    //  of every layer-1 section -- a global-memory round trip in the MFMA stream, 450 -> 950 cycles per section by the stamps)
    asm volatile("" : "+v"(x_mean), "+v"(x_std), "+v"(x_rstd));
    auto stage_x = [&]() {
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const double d = (double)xr[e] - x_mean;
            const double q0 = d * x_rstd;
            const double rr = fma(-q0, x_std, d);
            const double q1 = fma(rr, x_rstd, q0);
            xin[(xrow + e * (256 / KX)) * SX + xk] = (float)((rr == rr) ? q1 : q0);
        }
    };
    fetch_x(0);

    // ---- weights: registers, for the whole launch.  Host layout [member][wave][register / 4][lane][4]: register 4 kb + j of
    //      lane (column m = lane & 31, half hh) = W[gate(m) * H + unit(m)][8 kb + 4 hh + j] with m = gate * 8 + local unit
    //      Only layer 0's input columns (16 registers) are loaded here: step 0 of layer 0 needs nothing else (h_{-1} = 0), and its hand-over
    //      then travels while the other 384 registers come in (`load_weights`, called behind it).
    float w0[NW0];
    float w1[NW1];
    const f32x4* const s0 = reinterpret_cast<const f32x4*>(p.wcl[0]) + ((size_t)(member * 4 + wave) * (NW0 / 4)) * 64 + lane;
    const f32x4* const s1 = reinterpret_cast<const f32x4*>(p.wcl[1]) + ((size_t)(member * 4 + wave) * (NW1 / 4)) * 64 + lane;
#pragma unroll
    for (int i = 0; i < BX; ++i) {
        const f32x4 v = s0[i * 64];
        w0[4 * i] = v[0]; w0[4 * i + 1] = v[1]; w0[4 * i + 2] = v[2]; w0[4 * i + 3] = v[3];
    }
    auto load_weights = [&]() {
#pragma unroll
        for (int i = BX; i < NW0 / 4; ++i) {
            const f32x4 v = s0[i * 64];
            w0[4 * i] = v[0]; w0[4 * i + 1] = v[1]; w0[4 * i + 2] = v[2]; w0[4 * i + 3] = v[3];
        }
#pragma unroll
        for (int i = 0; i < NW1 / 4; ++i) {
            const f32x4 v = s1[i * 64];
            w1[4 * i] = v[0]; w1[4 * i + 1] = v[1]; w1[4 * i + 2] = v[2]; w1[4 * i + 3] = v[3];
        }
    };
    // accumulator start values (b_ih + b_hh) of this lane: registers 4 gate + j <-> unit member*32 + wave*8 + 4 hh + j; in LDS,
    // not in 32 more registers
#pragma unroll
    for (int l = 0; l < L; ++l)
#pragma unroll
        for (int gate = 0; gate < 4; ++gate) {
            f32x4 bv;
#pragma unroll
            for (int j = 0; j < 4; ++j) bv[j] = p.bias[l][gate * H + member * 32 + wave * 8 + 4 * hh + j];
            bias_s[((wave * L + l) * 4 + gate) * 64 + lane] = bv;
        }
    float cst[L][4];
#pragma unroll
    for (int l = 0; l < L; ++l)
#pragma unroll
        for (int j = 0; j < 4; ++j) cst[l][j] = 0.0f;

    // exchange buffer: descriptor for the compiler's stores, and the same words as a scalar tuple for the DMA asm
    const __amdgpu_buffer_rsrc_t hx_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.hx, 0, (int)p.hx_bytes, 0x00020000);
    const unsigned long long hx_addr = reinterpret_cast<unsigned long long>(p.hx);
    u32x4 hx_desc;
    hx_desc[0] = __builtin_amdgcn_readfirstlane((unsigned)hx_addr);
    hx_desc[1] = __builtin_amdgcn_readfirstlane((unsigned)(hx_addr >> 32) & 0xFFFFu);
    hx_desc[2] = __builtin_amdgcn_readfirstlane((unsigned)p.hx_bytes);
    hx_desc[3] = 0x00020000u;
    unsigned* const flags_of = p.xflags + (size_t)cluster * L * NFL;       // [layer][member*4 + wave] epoch = steps published
    // the looks at those flags (async_look.h): descriptor over the flag words, this cluster's byte offset, the lane's flag, the wave's zone
    const ape_desc_t fl_desc = ape_make_desc(p.xflags, (unsigned)((gridDim.x / GH) * L * NFL * sizeof(unsigned)));
    const unsigned fl_off = (unsigned)(cluster * L * NFL * sizeof(unsigned));
    const unsigned look_voff = (unsigned)((lane & (NFL - 1)) * sizeof(unsigned));
    const unsigned look_lds = (unsigned)reinterpret_cast<unsigned long long>(look_s) + (unsigned)(wave * 256);
    const unsigned* const look_mine = look_s + wave * 64 + lane;
    constexpr unsigned SET_BYTES = HL * sizeof(float);                     // one (layer, parity)
    auto hx_base = [&](int l, int par) -> unsigned { return (unsigned)((((size_t)cluster * L + l) * 2 + par) * SET_BYTES); };
    const unsigned hb0_lds = (unsigned)reinterpret_cast<unsigned long long>(hb0);     // LDS byte addresses
    const unsigned hb1_lds = (unsigned)reinterpret_cast<unsigned long long>(hb1);

    stage_x();
    if (T > 1) fetch_x(1);

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
    C32_TL(1);
        // hand-over form (ape_internal.h): write-through (`sc1`) payload stores unless the caller opted into the plain in-XCD form AND the
    // members were verified to share an XCD; uniform over the cluster
    const bool in_l2 = APE_HANDOVER_IN_L2(p.flags, ctl[3] != 0);

    // every wave polls for itself: have all member waves published epoch `want` of layer l?
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
    // a layer-step's 32 KB straight from the exchange buffer into its LDS block: wave w copies KiB w, w + 4, ... (8 of them);
    // piece k of that copy on its own, so that the copy can be spread over the k-blocks of a span
    const unsigned dma_voff = (unsigned)(lane * 16);
    // (blocking forms: the per-wave offset is laundered through an empty asm at every use, or hipcc hoists the sums of all pieces of all
    //  section forms out of the loop and runs out of scalar registers -- lstm_upper32.hip)
    auto opaque = [](unsigned v) -> unsigned { asm volatile("" : "+s"(v)); return v; };
    const unsigned wave_kib = (unsigned)(wave * 1024);
    auto issue_piece = [&](auto hot_tag, int l, int step, int k) {       // slices of layer l, step `step` -> hb0[step parity] / hb1
        constexpr bool HOT = decltype(hot_tag)::value;
        const unsigned wk = HOT ? wave_kib : opaque(wave_kib);
        const unsigned src = hx_base(l, step & 1) + wk;
        const unsigned dst = (l == 0 ? hb0_lds + (unsigned)((step & 1) * SET_BYTES) : hb1_lds) + wk;
        dma_1k<HOT>(dst, dma_voff, hx_desc, src, k * 4096);
    };
    constexpr std::true_type HOT{};
    constexpr std::false_type COLD{};
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

    // ---- gates + cell update, lane-local: registers 4 gate + j = gate of unit 4 hh + j, window n ---------------------------
    // (two cells at a time, the plain arithmetic on float2 values: v_pk_mul / v_pk_add / v_pk_fma_f32 do both cells in one issue
    //  slot -- VALU work is serial with the MFMAs, tools/experiments/mfma_chain_rate.hip; the transcendentals stay one by one)
    auto cell_update = [&](const f32x16& acc, float (&c_)[4], float (&hn)[4]) {
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
            const f32x2 c = fv * f32x2{c_[j], c_[j + 1]} + iv * gv;
            c_[j] = c[0]; c_[j + 1] = c[1];
            const f32x2 h = ov * (2.0f * rcp_2(1.0f + exp2_2(-2.885390081777927f * c)) - 1.0f);
            hn[j] = h[0]; hn[j + 1] = h[1];
        }
    };
    const unsigned pub_off = (unsigned)((((member * 4 + wave) * MR + n) * 8 + 4 * hh) * sizeof(float));    // this lane's 16 bytes of a slice set

    // ---- step 0 of layer 0 IN FRONT of the bulk of the weights: its product is the input span alone (4 k-blocks, 16 weight registers),
    //      and its hand-over -- store, acknowledgement, flag, the peers' flags -- would otherwise be the first thing the launch waits
    //      for with nothing to do (round 4: ~3 us of the 89 us launch at T = 6).  Phase 0's layer-1 section is idle anyway; the loop
    //      below starts at phase 1.
    if constexpr (!ENDS) {
        load_weights();
    } else {
        f32x16 acc;
#pragma unroll
        for (int gate = 0; gate < 4; ++gate) {
            const f32x4 bv = bias_s[((wave * L + 0) * 4 + gate) * 64 + lane];
            acc[4 * gate] = bv[0]; acc[4 * gate + 1] = bv[1]; acc[4 * gate + 2] = bv[2]; acc[4 * gate + 3] = bv[3];
        }
        span32<BX, false, NW0>(acc, xin + n * SX + hh * 4, 8, w0, 0, [&](int) {});
        mfma_drain(acc);
        float h0[4];
        cell_update(acc, cst[0], h0);
        const u32x4 hv = {__builtin_bit_cast(unsigned, h0[0]), __builtin_bit_cast(unsigned, h0[1]),
                          __builtin_bit_cast(unsigned, h0[2]), __builtin_bit_cast(unsigned, h0[3])};
        if (in_l2) __builtin_amdgcn_raw_buffer_store_b128(hv, hx_rsrc, hx_base(0, 0) + pub_off, 0, 0);
        else __builtin_amdgcn_raw_buffer_store_b128(hv, hx_rsrc, hx_base(0, 0) + pub_off, 0, 16 /* sc1: write-through */);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(flags_of + member * 4 + wave, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("" ::: "memory");
        load_weights();
        bar();                                                    // every wave is through with x_0 in LDS
        if (1 < T) {
            stage_x();
            if (2 < T) fetch_x(2);
        }
    }
    C32_TL(2);

    // Section (ph, l) = layer l on step t = ph - l.  It reads x_t / h^{l-1}_t and h^l_{t-1}, all published in phase ph - 1,
    // i.e. at least two sections ago; the ONE slice set it is still missing in LDS -- layer 0: h^0_{t-1} (published by the
    // last layer-0 section), layer 1: h^1_{t-1} (by the last layer-1 section) -- is prefetched by the section in front.
    // Per wave the vector-memory queue of a steady-state section is, in issue order:
    //   [publish store of the section in front]  flag look (1 load)  [x fetch, layer 1 only]  gather DMA for the NEXT section (8)
    //   publish store (1)
    // so at the top of a section everything but the youngest entry is waited for (`vmcnt(1)`: the slices are in LDS); behind the
    // barrier the store itself is waited for and its flag goes up.
    const int P = T + L - 1;                 // phases 0 .. P-1 compute; "phase" P: the final gather for the head
    bool prefetched = false;
#ifdef APE_C32_COUNTS
    // light counters of a product-like build (tests/tools/counts_c32.py): per workgroup (wave 0), per layer -- sections whose gather took the
    // blocking form / was prefetched, shader clocks between a section's entry and the exit of its top barrier
    unsigned lc_block[2] = {0u, 0u}, lc_go[2] = {0u, 0u}, lc_n[2] = {0u, 0u};
    unsigned long long lc_top[2] = {0ull, 0ull}, lc_t0 = 0ull;
    const unsigned long long lc_start = __builtin_amdgcn_s_memtime();
#endif
#ifdef APE_CLUSTER_STAMPS
    unsigned long long dg_block[2] = {0, 0}, dg_go[2] = {0, 0};      // diagnostic counters per layer (cluster 0, member 0)
    unsigned long long dgh[2][6] = {};       // ... and inside the spans: store drain + flag, x staging, judge, gather issue, (unused), eight hook-free blocks
    unsigned long long dgc[2][5] = {};       // steady-state sections: cycles in the top wait, the barrier, the MFMA spans, everything behind the barrier; count
    auto now = [&]() -> unsigned long long {
        const unsigned long long c = __builtin_readcyclecounter();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        return c;
    };
#endif
    // Two forms for the ends of a launch, where a section has no section in front whose MFMA stream could carry its exchange (round 4:
    // at the deployed T = 6 four of the launch's exchanges were exposed, ~3 us each of an 89 us launch):
    //   MODE 1  layer 1 on step 0 runs its input span alone (h_{-1} = 0): 32 k-blocks, so the look at the next section's flags and
    //           its gather sit at blocks 16 / 20 .. 27 instead of 30 / 34 .. 41, which it never reaches;
    // and ONE form for every layer-1 section with a recurrent span (round 5; round 4 had it for the last step of a short window only):
    //   MODE 3  layer 1 on step t >= 1 gathers its OWN missing slice set h^1_{t-1} under its input span: that span reads h^0_t, in LDS since
    //           the layer-0 section in front, so the section starts without a wait and WITHOUT a barrier; the look at the own layer's flags
    //           sits at block QF + 1 -- a whole layer-0 section (>= 4 us) behind the publish, where round 4's look from inside that
    //           layer-0 section came 1.4 us behind the flag and, with write-through stores, found it down in one section of ten
    //           (6.5 blocking gathers per 1024 x 64 launch and member against 1.0 with plain stores; tests/tools/counts_c32.py) --
    //           the eight pieces follow it, and the wait for them and the section's ONE barrier sit between the spans
    //           (lstm_cluster16.hip's place for it): 16+ k-blocks for the copies to land in instead of 8.  x staging moves behind that
    //           barrier; layer-0 sections no longer carry a look or a gather.  Outside the steady state (the last step of a short
    //           window sits behind an idle, i.e. empty, layer-0 section) a second look follows at block QF + 13; a wave whose looks
    //           failed blocks in front of the barrier.
    auto section = [&](auto steady_tag, auto layer_tag, auto mode_tag, const int ph) -> bool {
        constexpr bool ST = decltype(steady_tag)::value;          // steady state: 1 <= t <= T - 2 for both layers
        constexpr int l = decltype(layer_tag)::value;
        constexpr int MODE = decltype(mode_tag)::value;
        static_assert(MODE == 0 || (l == L - 1 && (MODE == 3 || !ST)), "the special forms are layer 1's");
        const int t = ph - l;
        const bool active = ST || (t >= 0 && t < T);
        // h^l_{t-1} exists and somebody reads it from here on (layer 0 at t == T: no layer-0 step any more, but layer 1's
        // last step takes h^0_{T-1} as its input)
        const bool need = ST || (t >= 1 && t <= T);
#ifdef APE_CLUSTER_STAMPS
        const unsigned long long c0 = ST ? now() : 0ull;
#endif
#ifdef APE_C32_COUNTS
        lc_t0 = __builtin_amdgcn_s_memtime();
#endif
        // ---- S0: this layer's slices of its last step into LDS ------------------------------------------------------------
        if (MODE == 3) {                                          // (the slices come in under the input span: nothing to wait for here)
        } else if (need) {
            if (!prefetched) {                                    // pipeline fill, a late peer, the final gather
#ifdef APE_CLUSTER_STAMPS
                dg_block[l] += 1;
#endif
#ifdef APE_C32_COUNTS
                lc_block[l] += 1u;
#endif
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                raise_pending();
                wait_flags(l, (unsigned)t);
#pragma unroll
                for (int k = 0; k < NDMA; ++k) issue_piece(COLD, l, t - 1, k);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(1)" ::: "memory");  // the prefetched copy; only the publish store is younger
            }
        }
        prefetched = false;
#ifdef APE_CLUSTER_STAMPS
        const unsigned long long c1 = ST ? now() : 0ull;
#endif
        if constexpr (MODE != 3) bar();
#ifdef APE_CLUSTER_STAMPS
        const unsigned long long c2 = ST ? now() : 0ull;
#endif
#ifdef APE_C32_COUNTS
        if (ST && MODE != 3) { lc_top[l] += __builtin_amdgcn_s_memtime() - lc_t0; lc_n[l] += 1u; }
#endif
        const int abort_word = (MODE == 3) ? 0 : ctl[0];          // (MODE 3 looks behind its own barrier, between the spans)
        C32_TL(8 + 2 * (2 * ph + l));
        // the next section: layer ln on step tn = its phase - ln; the slice set it is missing is h^{ln}_{tn-1}, epoch tn
        constexpr int ln = (l + 1 < L) ? l + 1 : 0;
        const int tn = (l + 1 < L) ? t - 1 : t + L;
        // (layer 1 gathers for itself, MODE 3: only layer-1 sections prefetch, for the layer-0 section behind them)
        const bool pre = (l == L - 1) && (ST || (tn >= 1 && tn <= T));
        unsigned peek = (unsigned)tn;
        bool go = false;
        // exchange work hung into the MFMA stream (k-block q of the whole layer-step, a constant after unrolling):
        //   QF  the flag owed for the OTHER layer's publish store, issued at the end of the section in front: a few blocks
        //       in that store has drained (`vmcnt(0)` with nothing else in the queue) and the peers, who look half-way
        //       through THEIR next section, are in no hurry
        //   QP  look at the flags the next section needs (one load per lane)    QJ  judge
        //   QJ .. QJ+7  one piece of the next section's gather per block
        constexpr int NBL = (l == 0) ? BX + BH : 2 * BH;
        // (positions swept on MI355X, 1024 x 64: QF 1 / 2 / 3 / 10 / 16 -> 826 / 833 / 819 / 828 / 858 us -- earlier stalls on the store's
        //  acknowledgement, later the peers' look finds nothing; look 12 blocks ahead of the judge instead of 4 -> 840: the flags are not up yet)
        constexpr int QF = C32_QF, QP = (MODE == 1) ? 16 : NBL / 2 - 2 + C32_QPD, QJ = (MODE == 1) ? 20 : NBL / 2 + 2 + C32_QPD;
        bool staged = false;
#ifdef APE_CLUSTER_STAMPS
        unsigned long long hk0 = 0, hk_free = 0;
#define HK_BEGIN() do { if (ST) hk0 = now(); } while (0)
#define HK_END(k) do { if (ST) dgh[l][k] += now() - hk0; } while (0)
#else
#define HK_BEGIN() do {} while (0)
#define HK_END(k) do {} while (0)
#endif
        auto mid = [&](int q) {
#ifdef APE_CLUSTER_STAMPS
            // eight k-blocks without any hook (after QF + 1, before QP): what does a bare block cost inside the real kernel?
            if (ST && q == QF + 2) hk_free = now();
            if (ST && q == QF + 10) dgh[l][5] += now() - hk_free;
#endif
            if (q == QF) {
                HK_BEGIN();
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                raise_pending();
                HK_END(0);
            }
            // x of the next layer-0 step: registers -> LDS (layer 0's readers of xin finished before this section's barrier), and
            // the fetch of the step after it EARLY in the section: issued at its end the loads were the youngest entries but one of
            // the memory queue, and the counted wait at the top of the next section sat out their whole latency (13 us per launch)
            if (l == L - 1 && q == ((MODE == 3) ? BH + 1 : QF + 1) && (ST || ph + 1 < T)) {
                HK_BEGIN();
                stage_x();
                if (ST || ph + 2 < T) fetch_x(ph + 2);
                staged = true;
                HK_END(1);
            }
            if constexpr (l == L - 1) {
                if (q == QP) look_issue(look_lds, look_voff, fl_desc, fl_off + (unsigned)(ln * NFL * sizeof(unsigned)));     // (always: no branch around it)
                if (q == QJ - 1) {                                // (the ds_read's latency passes under the next k-block)
                    look_landed();
                    peek = *look_mine;
                }
                if (q == QJ) {
                    HK_BEGIN();
                    go = pre && __all((int)(peek >= (unsigned)tn)) != 0;
                    HK_END(2);
                }
                if (q >= QJ && q < QJ + NDMA && go) {
                    HK_BEGIN();
                    issue_piece(HOT, ln, tn - 1, q - QJ);
                    HK_END(3);
                }
            }
        };
        float hnew[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (active) {
            // ---- stacked-gate product: one dependent chain of 32x32x2 MFMAs ---------------------------------------------------
            f32x16 acc;
#pragma unroll
            for (int gate = 0; gate < 4; ++gate) {
                const f32x4 bv = bias_s[((wave * L + l) * 4 + gate) * 64 + lane];
                acc[4 * gate] = bv[0]; acc[4 * gate + 1] = bv[1]; acc[4 * gate + 2] = bv[2]; acc[4 * gate + 3] = bv[3];
            }
            const int frag = n * 8 + hh * 4;                      // this lane's 16 bytes inside a [window][8 units] block
            if constexpr (l == 0) {
                span32<BX, false, NW0>(acc, xin + n * SX + hh * 4, 8, w0, 0, [&](int q) { mid(q); });
                if (ST || t > 0) span32<BH, false, NW0>(acc, hb0 + ((t - 1) & 1) * HL + frag, MR * 8, w0, 4 * BX, [&](int q) { mid(BX + q); });
            } else if constexpr (MODE == 3) {
                // input span: the common hooks (flag owed at QF, the look for the NEXT section at QP) + the look at the OWN layer's flags at
                // QO, the judge four blocks on, the gather behind it; outside the steady state a second look twelve blocks after the first
                constexpr int QO = QF + 1, QO2 = QO + 12;
                static_assert(QO2 + 4 + NDMA <= QP && QP < BH, "the own gather's hooks sit in front of the look for the next section");
                unsigned pk = (unsigned)t;
                bool got1 = false, got2 = false;
                span32<BH, true, NW1>(acc, hb0 + (t & 1) * HL + frag, MR * 8, w1, 0, [&](int q) {
                    mid(q);
                    if (q == QO) look_issue(look_lds, look_voff, fl_desc, fl_off + (unsigned)(l * NFL * sizeof(unsigned)));
                    if (q == QO + 3) {
                        look_landed();
                        pk = *look_mine;
                    }
                    if (q == QO + 4) got1 = __all((int)(pk >= (unsigned)t)) != 0;
                    if (q >= QO + 4 && q < QO + 4 + NDMA && got1) issue_piece(HOT, l, t - 1, q - (QO + 4));
                    if constexpr (!ST) {
                        if (q == QO2 && !got1) look_issue(look_lds, look_voff, fl_desc, fl_off + (unsigned)(l * NFL * sizeof(unsigned)));
                        if (q == QO2 + 3 && !got1) {
                            look_landed();
                            pk = *look_mine;
                        }
                        if (q == QO2 + 4 && !got1) got2 = __all((int)(pk >= (unsigned)t)) != 0;
                        if (q >= QO2 + 4 && q < QO2 + 4 + NDMA && got2) issue_piece(HOT, l, t - 1, q - (QO2 + 4));
                    }
                });
#ifdef APE_C32_COUNTS
                lc_t0 = __builtin_amdgcn_s_memtime();
#endif
                if (!got1 && !got2) {
#ifdef APE_CLUSTER_STAMPS
                    dg_block[l] += 1;
#endif
#ifdef APE_C32_COUNTS
                    lc_block[l] += 1u;
#endif
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    raise_pending();                              // (a section entered with its flag still owed: T = 2's only one)
                    wait_flags(l, (unsigned)t);
#pragma unroll
                    for (int k = 0; k < NDMA; ++k) issue_piece(COLD, l, t - 1, k);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                bar();                                            // every wave's pieces of h^1_{t-1} are in LDS; xin's readers are through
#ifdef APE_C32_COUNTS
                if (ST) { lc_top[l] += __builtin_amdgcn_s_memtime() - lc_t0; lc_n[l] += 1u; }
#endif
                if (ctl[0] != 0) return false;
                span32<BH, true, NW1>(acc, hb1 + frag, MR * 8, w1, 4 * BH, [&](int q) { mid(BH + q); });
            } else {
                span32<BH, true, NW1>(acc, hb0 + (t & 1) * HL + frag, MR * 8, w1, 0, [&](int q) { mid(q); });
                if (ST || t > 0) span32<BH, true, NW1>(acc, hb1 + frag, MR * 8, w1, 4 * BH, [&](int q) { mid(BH + q); });
                // (a section without a recurrent span -- step 0 of the long-window instantiation -- issues its look at block QP and never
                //  reaches the judge at QJ.  With a register destination that look stayed in flight over whatever hipcc gave the register to
                //  next: 2-4 % of the launches beside a memory-bound kernel returned a 32-window cluster off by 1e-2 at step 0 of layer 1,
                //  5e-4 at the end of a 9-step window, nothing visible from 24 steps on.  In its landing zone it harms nobody.)
            }
#ifdef APE_CLUSTER_STAMPS
            const unsigned long long c3 = ST ? now() : 0ull;
            if (ST) { dgc[l][0] += c1 - c0; dgc[l][1] += c2 - c1; dgc[l][2] += c3 - c2; }
#endif
            mfma_drain(acc);
            cell_update(acc, cst[l], hnew);
        }
        if (abort_word != 0) return false;                        // (a wave of this workgroup gave up in a blocking wait)
        if (!ST && pend_idx >= 0) {                               // (a section too short to reach block QF, or an idle one)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            raise_pending();
        }
        // (first step of a layer, idle sections: the hooks did not run, the next section gathers in the blocking form)
        if (go) prefetched = true;
#ifdef APE_CLUSTER_STAMPS
        if (go) dg_go[ln] += 1;
#endif
#ifdef APE_C32_COUNTS
        if (go) lc_go[ln] += 1u;
#endif
        if constexpr (l == L - 1) {
            // (an idle section, whose hooks did not run: the same here)
            if (!ST && !staged && ph + 1 < T) {
                stage_x();
                if (ph + 2 < T) fetch_x(ph + 2);
            }
        }
        // ---- publish: this lane's four fresh h values are one 16-byte piece of the exchange layout ----------------------------
        //      (exactly ONE store instruction per wave and section: the counted wait at the top of the next section relies on it)
        {
            const u32x4 hv = {__builtin_bit_cast(unsigned, hnew[0]), __builtin_bit_cast(unsigned, hnew[1]),
                              __builtin_bit_cast(unsigned, hnew[2]), __builtin_bit_cast(unsigned, hnew[3])};
            const unsigned off = active ? hx_base(l, t & 1) + pub_off : 0x80000000u;
            if (in_l2) __builtin_amdgcn_raw_buffer_store_b128(hv, hx_rsrc, off, 0, 0);
            else __builtin_amdgcn_raw_buffer_store_b128(hv, hx_rsrc, off, 0, 16 /* sc1: write-through */);
            if (active) {
                pend_idx = l * NFL + member * 4 + wave;
                pend_epoch = (unsigned)(t + 1);
            }
        }
#ifdef APE_CLUSTER_STAMPS
        if (ST) { dgc[l][3] += now() - c2; dgc[l][4] += 1; }
#endif
        C32_TL(9 + 2 * (2 * ph + l));
        return true;
    };
    bool ok = true;
#pragma unroll 1
    for (int ph = ENDS ? 1 : 0; ph < P && ok; ++ph) {               // (ENDS: phase 0 = step 0 of layer 0, done above)
        // steady state: both layers active with a recurrent span, a next section to prefetch for, x to stage and to fetch
        const bool st0 = ph >= 2 && ph <= T - 3, st1 = st0;
        using M0 = std::integral_constant<int, 0>;
        using L0 = std::integral_constant<int, 0>;
        using L1 = std::integral_constant<int, 1>;
        ok = st0 ? section(std::true_type{}, L0{}, M0{}, ph) : section(std::false_type{}, L0{}, M0{}, ph);
        if (!ok) break;
        const int t1 = ph - 1;
        bool done1 = false;
        if constexpr (ENDS) {
            if (!st1 && t1 == 0 && T > 1) {
                ok = section(std::false_type{}, L1{}, std::integral_constant<int, 1>{}, ph);
                done1 = true;
            }
        }
        // (a step with a recurrent span: the section gathers h^1_{t-1} for itself, MODE 3; idle sections and step 0 of the long-window
        //  instantiation: the plain form)
        using M3 = std::integral_constant<int, 3>;
        if (!done1 && t1 >= 1 && t1 < T) {
            ok = st1 ? section(std::true_type{}, L1{}, M3{}, ph) : section(std::false_type{}, L1{}, M3{}, ph);
            done1 = true;
        }
        if (!done1) ok = section(std::false_type{}, L1{}, M0{}, ph);
    }
    if (!ok) return;
    // ---- final gather: h^{L-1}_{T-1} of every member (a section of layer L-1 "on step T": S0 only) -------------------------------
    {
        // the last step's flags go up per MEMBER: every wave has drained its store in front of the barrier, wave 0 stores the member's four
        // words in one instruction.  (Round 6: a cluster's 32 per-wave flag stores to one cache line, all issued at about the same time,
        // complete one after the other on the memory side -- the last one turned visible 3 .. 7 us after its issue, measured on
        // lstm_cluster16.hip's final gather, profiles/r06_flag_serialisation.md.  In the sections the looks are asynchronous and the
        // stores spread out; here every wave of the cluster is waiting for exactly these words.  The arrival-counter form of the other
        // kernels -- the last of a member's waves to drain raises for all four -- was tried in the sections of the short-window instantiation
        // and DEADLOCKS here: with two layers (and MODE 3's own gather) a wave on the blocking path waits, in front of the section's barrier, for
        // flags that include its OWN member's newest ones, which then need the arrival of waves that wait for it behind that barrier.  In
        // lstm_cluster.hip / lstm_cluster16.hip a blocking wait is only ever for flags at least a section older than the member's own pending
        // raise, and every wave of the member has passed that raise.)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        bar();
        if (pend_idx >= 0) {
            if (wave == 0 && lane < 4)
                __hip_atomic_store(flags_of + (pend_idx - wave) + lane, pend_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            pend_idx = -1;
        }
        if (!prefetched) {
            wait_flags(L - 1, (unsigned)T);
#pragma unroll
            for (int k = 0; k < NDMA; ++k) issue_piece(COLD, L - 1, T - 1, k);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        bar();
        if (ctl[0] != 0) return;
        C32_TL(4);
    }

#ifdef APE_CLUSTER_STAMPS
    if (p.dbg_wg != nullptr && tid == 0 && cluster == 0 && member == 0)
        for (int k = 0; k < 2; ++k) { p.dbg_wg[16 + k] = dg_block[k]; p.dbg_wg[18 + k] = dg_go[k]; }
    if (p.dbg_wg != nullptr && lane == 0 && cluster == 0 && member == 0)
        for (int l = 0; l < 2; ++l)
            for (int k = 0; k < 5; ++k) p.dbg_wg[32 + wave * 16 + l * 5 + k] = dgc[l][k];
    if (p.dbg_wg != nullptr && lane == 0 && cluster == 0 && member == 0)
        for (int l = 0; l < 2; ++l)
            for (int k = 0; k < 6; ++k) p.dbg_wg[128 + wave * 16 + l * 6 + k] = dgh[l][k];
#endif
#ifdef APE_C32_COUNTS
    if (p.dbg_wg != nullptr && tid == 0 && cluster < 24) {
        unsigned long long* o = p.dbg_wg + 512 + (cluster * GH + member) * 8;
        o[0] = lc_block[0]; o[1] = lc_block[1]; o[2] = lc_go[0]; o[3] = lc_go[1];
        o[4] = lc_top[0]; o[5] = lc_top[1]; o[6] = lc_n[0]; o[7] = __builtin_amdgcn_s_memtime() - lc_start;
    }
#endif
    // ---- head: member m finishes windows 4m .. 4m+3 of the cluster's 32 ----------------------------------------------------------------
    {
        constexpr int RPM = MR / GH;
        // (window, target) dot products over H, 4 lanes each (k-blocks interleaved by 4), combined by lane shuffles
        const int part = tid & 3;
        for (int oi = tid >> 2; oi < ((RPM * O + 63) / 64) * 64; oi += 64) {
            const bool live = oi < RPM * O;
            const int rr = live ? oi / O : 0, o = live ? oi - rr * O : 0;
            const int row = member * RPM + rr, b = row0 + row;
            float s_acc = 0.0f;
            if (live) {
                const float* wv = p.w_out + (size_t)o * H;
                // unit k of window `row` lives at hb1[(k / 8) * (MR * 8) + row * 8 + (k % 8)]
                for (int kb = part; kb < H / 8; kb += 4) {
                    const f32x4 a0 = *reinterpret_cast<const f32x4*>(hb1 + kb * MR * 8 + row * 8);
                    const f32x4 a1 = *reinterpret_cast<const f32x4*>(hb1 + kb * MR * 8 + row * 8 + 4);
                    const f32x4 u0 = *reinterpret_cast<const f32x4*>(wv + kb * 8), u1 = *reinterpret_cast<const f32x4*>(wv + kb * 8 + 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) s_acc = fmaf(a0[j], u0[j], s_acc);
#pragma unroll
                    for (int j = 0; j < 4; ++j) s_acc = fmaf(a1[j], u1[j], s_acc);
                }
            }
            s_acc += __shfl_xor(s_acc, 1, 64);
            s_acc += __shfl_xor(s_acc, 2, 64);
            if (p.y != nullptr && live && part == 0 && b < p.B) p.y[(size_t)b * O + o] = s_acc + p.b_out[o];
        }
    }
    C32_TL(5);
    // ---- self-cleaning: the last workgroup out re-zeroes every polled word ------------------------------------------
    __syncthreads();
    if (tid == 0)
        ctl[2] = (__hip_atomic_fetch_add(p.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) ? 1 : 0;
    __syncthreads();
    if (ctl[2] != 0) {
        const int n_flags = (int)(gridDim.x / GH) * L * NFL;
        for (int i = tid; i < n_flags; i += 256) __hip_atomic_store(p.xflags + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int i = tid; i < (int)gridDim.x; i += 256) __hip_atomic_store(xcc_words + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid < 8) __hip_atomic_store(class_ticket + tid * 16, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid == 0) __hip_atomic_store(p.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

constexpr size_t smem_bytes32() {
    return ((size_t)3 * 8 * 4 * 32 * 8 + (size_t)32 * 36) * sizeof(float) + (size_t)4 * 2 * 4 * 64 * 16 + (size_t)4 * 64 * sizeof(unsigned) + 16;
}

}  // namespace

bool ape_cluster32_supported(int H, int L, int KX) { return H == 256 && L == 2 && KX == 32; }

hipError_t ape_prepare_lstm_cluster32(int H, int L, int KX) {
    if (!ape_cluster32_supported(H, L, KX)) return hipSuccess;
    static_assert(smem_bytes32() <= APE_LDS_BYTES, "LDS layout exceeds a CU");
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ape_lstm_cluster32<256, 2, 32, false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, APE_LDS_BYTES);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&ape_lstm_cluster32<256, 2, 32, true>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, APE_LDS_BYTES);
}

// `clusters` = 32-window clusters needed; the grid is rounded up to whole block-index classes (8 clusters), the extra
// clusters own windows past the batch and only take part in the formation
hipError_t ape_launch_lstm_cluster32(int H, int L, int KX, int clusters, const ClusterParams& p, hipStream_t stream) {
    if (!ape_cluster32_supported(H, L, KX)) return hipErrorInvalidValue;
    const int grid_clusters = (clusters + 7) / 8 * 8;
    constexpr size_t smem = smem_bytes32();
    if (p.T <= APE_C32_ENDS_MAX_T) hipLaunchKernelGGL((ape_lstm_cluster32<256, 2, 32, true>), dim3(grid_clusters * 8), dim3(256), smem, stream, p);
    else hipLaunchKernelGGL((ape_lstm_cluster32<256, 2, 32, false>), dim3(grid_clusters * 8), dim3(256), smem, stream, p);
    return hipGetLastError();
}
