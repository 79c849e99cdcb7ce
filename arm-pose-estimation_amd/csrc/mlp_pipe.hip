// MLP regressor (DropoutFF, reference estimate/nn_models.py:313-370: Linear(I,256) + leaky_relu, 2 x [Linear(256,256) + leaky_relu],
// Linear(256,O)) for batches that fill the chip, eval mode: a weight-stationary TWO-STAGE PIPELINE over pairs of CUs.
//
// mlp_tile16.hip streams every layer's weights from L2 for every 16/32-row tile (52 % of the f32 MFMA peak at 262 144 rows).
// Here the weights never move: of a pair of workgroups (one per CU, the pair inside one XCD where the dispatcher allows it --
// membership by arrival ticket within the block-index class, as lstm_cluster32.hip) workgroup A keeps the input layer and the
// first hidden layer in its registers (288 per lane), workgroup B the second hidden layer and the output layer (288), and 32-row
// tiles flow A -> B through a four-slot ring in memory (32 KB per tile, in the MFMA fragment order of the consumer, copied into
// LDS by LDS-DMA).  There is no round trip anywhere -- an MLP has no recurrence -- so the hand-over latency disappears behind
// the ring's depth; each stage is 288 v_mfma_f32_32x32x2_f32 per wave and tile (two 32-unit column tiles per wave: two
// independent accumulator chains), separated by two (A) / three (B) workgroup barriers.
//
// Same arithmetic as mlp_tile16.hip up to float32 summation order.  Eval mode, last-step rows, H = 256, two hidden layers,
// I <= 32, O <= 32; everything else stays on mlp_tile16.hip.
#include "ape_internal.h"
#include "../../include/ape_hip.h"
#include <cstdlib>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((__vector_size__(4 * sizeof(unsigned))));
constexpr unsigned SPIN_LIMIT = 1u << 22;
constexpr int H = 256, KX = 32, TR = 32;            // hidden width, padded input width, rows per tile
constexpr int NSLOT = 4;                            // ring slots per pair
constexpr int HLF = (H / 8) * TR * 8;               // floats of one tile of activations [k-block 32][row 32][8] = 32 KB
constexpr int SX = KX + 4;

template <bool AG>
__device__ __forceinline__ void mfma32(f32x16& acc, float w, float a) {
    if constexpr (AG) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "a"(w), "v"(a));
    else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "v"(a));
}
__device__ __forceinline__ void mfma_drain2(f32x16& a, f32x16& b) { asm volatile("s_nop 15\n\ts_nop 7" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void mfma_drain1(f32x16& a) { asm volatile("s_nop 15\n\ts_nop 7" : "+v"(a)); }
__device__ __forceinline__ void dma_1k(unsigned lds_addr, unsigned voff, u32x4 rsrc, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen sc1 lds"
                 :: "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

// NB k-blocks of 8 for two column tiles at once: acc0 / acc1 += W (registers w[w0 + ct * NWT + ...]) x activations (LDS, one
// ds_read_b128 per block feeds the eight MFMAs of both tiles), fragments fetched two blocks ahead
struct NoHook { __device__ __forceinline__ void operator()(int) const {} };
template <int NB, bool AG, int NW, typename Hook = NoHook>
__device__ __forceinline__ void span2(f32x16& acc0, f32x16& acc1, const float* __restrict__ src, int stride, const float (&w)[NW], int w0, int nwt,
                                      Hook hook = Hook()) {
    f32x4 a0 = *reinterpret_cast<const f32x4*>(src);
    f32x4 a1 = (NB > 1) ? *reinterpret_cast<const f32x4*>(src + stride) : a0;
    f32x4 a2 = a1;
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
        if (kb + 2 < NB) a2 = *reinterpret_cast<const f32x4*>(src + stride * (kb + 2));
        hook(kb);                                          // (memory operations of the hand-over, in the shadow of the MFMAs)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            mfma32<AG>(acc0, w[w0 + 4 * kb + j], a0[j]);
            mfma32<AG>(acc1, w[w0 + nwt + 4 * kb + j], a0[j]);
        }
        a0 = a1;
        a1 = a2;
    }
}

struct PipeParams {
    MlpParams m;
    const float* wa0;      // stage A, input layer:   [wave 4][ct 2][kb 4][lane 64][4]
    const float* wa1;      // stage A, hidden layer 1: [wave 4][ct 2][kb 32][lane 64][4]
    const float* wb2;      // stage B, hidden layer 2: same shape
    const float* wbo;      // stage B, output layer:   [wave 4][kb 8][lane 64][4]  (W_out[o = lane & 31][64 w + 8 kb + 4 hh + j], 0 for o >= O)
    float* ring;           // [pair][slot 4][HLF]
    size_t ring_bytes;
    unsigned* ctl;         // [8 class tickets x 16][status][done][pad..][pair][full 4 x 4 | empty 4 x 4]
    unsigned diag;         // timing experiments (APE_PIPE_DIAG; results are garbage): 1 = never wait for the peer, 2 = stage A only, 4 = stage B only, 8 = no ring stores, 16 = no leaky_relu, 32 = no barriers, 64 = no h0 / h2 writes
};

__global__ __launch_bounds__(256, 1) void ape_mlp_pipe(const PipeParams pp) {
    const MlpParams& p = pp.m;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, hh = lane >> 5;
    const int frag = n * 8 + hh * 4;
    const float slope = p.neg_slope;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int* ctl_s = reinterpret_cast<int*>(smem);                 // [0] abort, [1] ticket
    float* lds = smem + 16;

    unsigned* const class_ticket = pp.ctl;
    unsigned* const status = pp.ctl + 8 * 16;
    unsigned* const done = status + 1;
    const int cls = blockIdx.x & 7;
    if (tid == 0) {
        ctl_s[0] = 0;
        ctl_s[1] = -1;
        if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
            const unsigned tk = __hip_atomic_fetch_add(class_ticket + cls * 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (tk < gridDim.x / 8) ctl_s[1] = (int)tk;
            else __hip_atomic_store(status, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    if (ctl_s[1] < 0) return;
    const int ticket = __builtin_amdgcn_readfirstlane(ctl_s[1]);
    const int pair = (ticket >> 1) * 8 + cls, role = ticket & 1;
    const int n_pairs = gridDim.x / 2;
    const int n_tiles = (p.N + TR - 1) / TR;
    unsigned* const full = pp.ctl + 256 + pair * 32;            // [slot][producer wave] = tiles of that slot written
    unsigned* const empty = full + 16;                          // [slot][consumer wave] = tiles of that slot copied out
    const __amdgpu_buffer_rsrc_t ring_rsrc = __builtin_amdgcn_make_buffer_rsrc(pp.ring, 0, (int)pp.ring_bytes, 0x00020000);
    const unsigned long long ring_addr = reinterpret_cast<unsigned long long>(pp.ring);
    u32x4 ring_desc;
    ring_desc[0] = __builtin_amdgcn_readfirstlane((unsigned)ring_addr);
    ring_desc[1] = __builtin_amdgcn_readfirstlane((unsigned)(ring_addr >> 32) & 0xFFFFu);
    ring_desc[2] = __builtin_amdgcn_readfirstlane((unsigned)pp.ring_bytes);
    ring_desc[3] = 0x00020000u;
    auto slot_base = [&](int slot) -> unsigned { return (unsigned)(((size_t)pair * NSLOT + slot) * HLF * sizeof(float)); };
    auto bar = [&]() { if (pp.diag & 32u) return; asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    // all 4 words of `f` (one per peer wave) have reached `want`
    auto wait_words = [&](const unsigned* f, unsigned want) {
        if (pp.diag & 1u) return;
        unsigned spins = 0;
        while (true) {
            unsigned v = want;
            if (lane < 4) v = __hip_atomic_load(f + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__all((int)(v >= want))) return;
            if (++spins > SPIN_LIMIT || ((spins & 255u) == 0u && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                if (lane == 0) {
                    ctl_s[0] = 1;
                    __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                return;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    };
    // the same wait in two halves, so that the L2 round trip of the look runs beside the MFMAs: the four words are fetched here ...
    auto poll_begin = [&](const unsigned* f) -> unsigned {
        unsigned v;
        const unsigned* a = f + (lane & 3);
        asm volatile("global_load_dword %0, %1, off sc1" : "=v"(v) : "v"(a) : "memory");
        return v;
    };
    // ... and looked at here (behind an s_waitcnt vmcnt(0)); only a word that is still behind goes into the polling loop
    auto poll_end = [&](unsigned v, const unsigned* f, unsigned want) {
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(v) :: "memory");
        if ((pp.diag & 1u) || __all((int)(v >= want))) return;
        wait_words(f, want);
    };
    auto leaky16 = [&](f32x16& a) {
        if (pp.diag & 16u) return;
#pragma unroll
        for (int i = 0; i < 16; ++i) {                                      // slope < 1: max(y, slope y) = leaky_relu / relu
            const float sa = slope * a[i];
            float r;
            asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a[i]), "v"(sa));             // (fmaxf adds a canonicalising v_max per element)
            a[i] = r;
        }
    };
    // this lane's 16 values of column tile ct = units 64 wave + 32 ct + 8 (i >> 2) + 4 hh + (i & 3) of row n: four 16-byte pieces of
    // the tile layout [unit / 8][row][8]
    auto tile_off = [&](int ct, int q) -> int { return ((8 * wave + 4 * ct + q) * TR + n) * 8 + 4 * hh; };

    if ((pp.diag & 2u) && role == 1) {
    } else if ((pp.diag & 4u) && role == 0) {
    } else if (role == 0) {
        // =========================== stage A: x -> layer 0 -> layer 1 -> ring ==========================================
        float* xin = lds;                                      // [TR][SX]
        float* h0 = xin + TR * SX;                             // [HLF]
        float* bias_s = h0 + HLF;                              // [2 layers][H]
        float w0[2 * 4 * 4];                                   // input layer: 16 registers per column tile
        float w1[2 * 32 * 4];                                  // hidden layer 1: 128 per column tile (accumulator file)
        {
            const f32x4* s0 = reinterpret_cast<const f32x4*>(pp.wa0) + (size_t)wave * 2 * 4 * 64 + lane;
#pragma unroll
            for (int i = 0; i < 8; ++i) { const f32x4 v = s0[i * 64]; w0[4 * i] = v[0]; w0[4 * i + 1] = v[1]; w0[4 * i + 2] = v[2]; w0[4 * i + 3] = v[3]; }
            const f32x4* s1 = reinterpret_cast<const f32x4*>(pp.wa1) + (size_t)wave * 2 * 32 * 64 + lane;
#pragma unroll
            for (int i = 0; i < 64; ++i) { const f32x4 v = s1[i * 64]; w1[4 * i] = v[0]; w1[4 * i + 1] = v[1]; w1[4 * i + 2] = v[2]; w1[4 * i + 3] = v[3]; }
            // every weight register "used" here: the compiler waits for the loads in front of the loop instead of carrying
            // s_waitcnt vmcnt(n) into its MFMA stream (where they would wait for this tile's ring stores as well)
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("" : "+v"(w0[i]));
#pragma unroll
            for (int i = 0; i < 256; ++i) asm volatile("" : "+a"(w1[i]));
        }
        bias_s[tid] = p.bias[0][tid];                          // (the loop's first barrier publishes them)
        bias_s[H + tid] = p.bias[1][tid];
        auto load_bias = [&](f32x16& acc, int l, int ct) {     // unit 64 wave + 32 ct + 8 q + 4 hh + j of accumulator register 4 q + j
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_s + l * H + 64 * wave + 32 * ct + 8 * q + 4 * hh);
                acc[4 * q] = bv[0]; acc[4 * q + 1] = bv[1]; acc[4 * q + 2] = bv[2]; acc[4 * q + 3] = bv[3];
            }
        };
        // x: thread owns 4 (row, k) elements of a tile, all with the same k; f64 z-score like mlp_tile16.hip
        const bool normalize = (p.flags & APE_FLAG_NORMALIZE_INPUT) != 0;
        const int xk = tid & 31, xrow = tid >> 5;
        const double x_mean = (normalize && xk < p.I) ? p.xx_m[xk] : 0.0;
        const double x_std = (normalize && xk < p.I) ? p.xx_s[xk] : 1.0;
        float xr[4];
        auto fetch_x = [&](int tile) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const long long row = (long long)tile * TR + xrow + 8 * e;
                xr[e] = (xk < p.I && row < p.N) ? p.x[(size_t)row * p.row_stride + p.row_offset + xk] : 0.0f;
            }
        };
        auto stage_x = [&]() {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                xin[(xrow + 8 * e) * SX + xk] = normalize ? (float)(((double)xr[e] - x_mean) / x_std) : xr[e];
        };
        int pend_slot = -1;
        unsigned pend_epoch = 0u;
        // x of this pair's first tile into LDS, the second tile's on its way
        if (pair < n_tiles) {
            fetch_x(pair);
            stage_x();
            if (pair + n_pairs < n_tiles) fetch_x(pair + n_pairs);
        }
        int it = 0;
        for (int tile = pair; tile < n_tiles; tile += n_pairs, ++it) {
            const int slot = it & (NSLOT - 1);
            bar();                                             // xin of this tile visible; every wave is done with layer 1 of the tile in front: h0 is free
            if (ctl_s[0] != 0) return;
            // has the consumer copied this tile's ring slot out (tile it - NSLOT of the pair)?  Asked here, looked at in layer 1
            unsigned slot_free = 0u;
            if (it >= NSLOT) slot_free = poll_begin(empty + slot * 4);
            f32x16 acc0, acc1;
            load_bias(acc0, 0, 0);
            load_bias(acc1, 0, 1);
            span2<4, false, 32>(acc0, acc1, xin + n * SX + hh * 4, 8, w0, 0, 16);
            mfma_drain2(acc0, acc1);
            leaky16(acc0);
            leaky16(acc1);
            if (!(pp.diag & 64u))
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                *reinterpret_cast<f32x4*>(h0 + tile_off(0, q)) = f32x4{acc0[4 * q], acc0[4 * q + 1], acc0[4 * q + 2], acc0[4 * q + 3]};
                *reinterpret_cast<f32x4*>(h0 + tile_off(1, q)) = f32x4{acc1[4 * q], acc1[4 * q + 1], acc1[4 * q + 2], acc1[4 * q + 3]};
            }
            bar();                                             // h0 complete; xin is free
            if (tile + n_pairs < n_tiles) {                    // the next tile's x (fetched a tile ago) into LDS, the one after it on its way
                stage_x();
                if (tile + 2 * n_pairs < n_tiles) fetch_x(tile + 2 * n_pairs);
            }
            load_bias(acc0, 1, 0);
            load_bias(acc1, 1, 1);
            // the flag owed for the tile in front is raised half-way through this layer: its ring stores went out 3.5 us ago and the
            // x fetch behind them is as old, so the wait costs nothing (at the top of the layer it would: stores take ~1 us)
            span2<32, true, 256>(acc0, acc1, h0 + frag, TR * 8, w1, 0, 128, [&](int kb) {
                if (kb == 16) {
                    if (it >= NSLOT) poll_end(slot_free, empty + slot * 4, (unsigned)(it / NSLOT));
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (pend_slot >= 0 && lane == 0)
                        __hip_atomic_store(full + pend_slot * 4 + wave, pend_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            });
            mfma_drain2(acc0, acc1);
            leaky16(acc0);
            leaky16(acc1);
            const unsigned base = slot_base(slot);
            if (!(pp.diag & 8u))
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                // (whole-vector casts: hipcc of ROCm 7.2 folds a vector built from per-element bit casts of an asm result into
                // four copies of its element 0)
                const u32x4 v0 = __builtin_bit_cast(u32x4, f32x4{acc0[4 * q], acc0[4 * q + 1], acc0[4 * q + 2], acc0[4 * q + 3]});
                const u32x4 v1 = __builtin_bit_cast(u32x4, f32x4{acc1[4 * q], acc1[4 * q + 1], acc1[4 * q + 2], acc1[4 * q + 3]});
                __builtin_amdgcn_raw_buffer_store_b128(v0, ring_rsrc, base + (unsigned)(tile_off(0, q) * 4), 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(v1, ring_rsrc, base + (unsigned)(tile_off(1, q) * 4), 0, 0);
            }
            pend_slot = slot;
            pend_epoch = (unsigned)(it / NSLOT + 1);
        }
        if (pend_slot >= 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_store(full + pend_slot * 4 + wave, pend_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else {
        // =========================== stage B: ring -> layer 2 -> output layer -> y ======================================
        float* inb = lds;                                      // [2][HLF]  h1 tiles, double-buffered
        float* h2 = inb + 2 * HLF;                             // [HLF]
        float* bias_s = h2 + HLF;                              // [2 ct][4 q][lane 64] f32x4
        float* pbuf = bias_s + H;                              // [wave 4][row 32][O] partial outputs (1024 floats per wave)
        float w2[2 * 32 * 4];
        float wo[8 * 4];
        {
            const f32x4* s2 = reinterpret_cast<const f32x4*>(pp.wb2) + (size_t)wave * 2 * 32 * 64 + lane;
#pragma unroll
            for (int i = 0; i < 64; ++i) { const f32x4 v = s2[i * 64]; w2[4 * i] = v[0]; w2[4 * i + 1] = v[1]; w2[4 * i + 2] = v[2]; w2[4 * i + 3] = v[3]; }
            const f32x4* so = reinterpret_cast<const f32x4*>(pp.wbo) + (size_t)wave * 8 * 64 + lane;
#pragma unroll
            for (int i = 0; i < 8; ++i) { const f32x4 v = so[i * 64]; wo[4 * i] = v[0]; wo[4 * i + 1] = v[1]; wo[4 * i + 2] = v[2]; wo[4 * i + 3] = v[3]; }
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("" : "+v"(wo[i]));
#pragma unroll
            for (int i = 0; i < 256; ++i) asm volatile("" : "+a"(w2[i]));
        }
        bias_s[tid] = p.bias[2][tid];                          // (the loop's first barrier publishes them)
        float bo[16];                                          // the output bias rides in wave 0's partial sum
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int o = 8 * (i >> 2) + 4 * hh + (i & 3);
            bo[i] = (wave == 0 && o < p.O) ? p.b_out[o] : 0.0f;
        }
        auto load_bias = [&](f32x16& acc, int ct) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_s + 64 * wave + 32 * ct + 8 * q + 4 * hh);
                acc[4 * q] = bv[0]; acc[4 * q + 1] = bv[1]; acc[4 * q + 2] = bv[2]; acc[4 * q + 3] = bv[3];
            }
        };
        const unsigned inb_lds = (unsigned)reinterpret_cast<unsigned long long>(inb);
        const unsigned dma_voff = (unsigned)(lane * 16);
        auto issue_copy = [&](int it_) {                       // tile it_ of this pair: ring slot -> inb[it_ & 1]; wave w copies KiB w, w + 4, ...
            const unsigned src = slot_base(it_ & (NSLOT - 1)) + (unsigned)(wave * 1024);
            const unsigned dst = inb_lds + (unsigned)((it_ & 1) * HLF * 4 + wave * 1024);
#pragma unroll
            for (int k = 0; k < 8; ++k) dma_1k(dst + (unsigned)(k * 4096), dma_voff, ring_desc, src + (unsigned)(k * 4096));
        };
        const int my_tiles = (pair < n_tiles) ? (n_tiles - pair + n_pairs - 1) / n_pairs : 0;
        if (my_tiles > 0) {
            wait_words(full + 0, 1u);
            issue_copy(0);
        }
        // y of one tile from its four partial sums; issued a tile late, in front of the next copy, so that the wait for the copy
        // at the top of the loop finds the y stores long gone
        auto write_y = [&](int tile) {                         // (a tile's 32 rows of y are one contiguous run of 32 O floats)
            const long long left = ((long long)p.N - (long long)tile * TR) * p.O;
            float* ydst = p.y + (size_t)tile * TR * p.O;
            for (int idx = tid; idx < TR * p.O; idx += 256)
                if (idx < left)
                    ydst[idx] = (pbuf[idx] + pbuf[32 * TR + idx]) + (pbuf[2 * 32 * TR + idx] + pbuf[3 * 32 * TR + idx]);
        };
        for (int it = 0; it < my_tiles; ++it) {
            const int slot = it & (NSLOT - 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's KiBs of tile `it` are in LDS (and its y stores are out)
            if (lane == 0) __hip_atomic_store(empty + slot * 4 + wave, (unsigned)(it / NSLOT + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bar();                                             // all KiBs of the tile in LDS; h2 of the tile in front is free, its partial sums are complete
            if (ctl_s[0] != 0) return;
            if (it > 0) write_y(pair + (it - 1) * n_pairs);   // (its partial sums were complete at the barrier above)
            unsigned next_full = 0u;
            f32x16 acc0, acc1;
            load_bias(acc0, 0);
            load_bias(acc1, 1);
            // the next tile's copy into the other buffer (its readers finished before the barrier above) starts a third into this
            // layer (the pair's producer raises that tile's flag half-way through the tile behind it); the look at the flag is two
            // k-blocks older
            span2<32, true, 256>(acc0, acc1, inb + (it & 1) * HLF + frag, TR * 8, w2, 0, 128, [&](int kb) {
                if (kb == 4 && it + 1 < my_tiles) next_full = poll_begin(full + ((it + 1) & (NSLOT - 1)) * 4);
                if (kb == 12 && it + 1 < my_tiles) {
                    poll_end(next_full, full + ((it + 1) & (NSLOT - 1)) * 4, (unsigned)((it + 1) / NSLOT + 1));
                    issue_copy(it + 1);
                }
            });
            mfma_drain2(acc0, acc1);
            leaky16(acc0);
            leaky16(acc1);
            if (!(pp.diag & 64u))
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                *reinterpret_cast<f32x4*>(h2 + tile_off(0, q)) = f32x4{acc0[4 * q], acc0[4 * q + 1], acc0[4 * q + 2], acc0[4 * q + 3]};
                *reinterpret_cast<f32x4*>(h2 + tile_off(1, q)) = f32x4{acc1[4 * q], acc1[4 * q + 1], acc1[4 * q + 2], acc1[4 * q + 3]};
            }
            bar();                                             // h2 complete; the partial sums of the tile in front have been read
            // output layer: this wave's K slice (units 64 wave .. +63 = k-blocks 8 wave .. +7) of all 32 (padded) outputs
            f32x16 acco;
#pragma unroll
            for (int i = 0; i < 16; ++i) acco[i] = bo[i];
            {
                const float* src = h2 + (8 * wave) * TR * 8 + frag;
#pragma unroll
                for (int kb = 0; kb < 8; ++kb) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(src + kb * TR * 8);
#pragma unroll
                    for (int j = 0; j < 4; ++j) mfma32<false>(acco, wo[4 * kb + j], a[j]);
                }
            }
            mfma_drain1(acco);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int o = 8 * (i >> 2) + 4 * hh + (i & 3);
                if (o < p.O) pbuf[wave * (32 * TR) + n * p.O + o] = acco[i];
            }
        }
        if (my_tiles > 0) {
            bar();
            write_y(pair + (my_tiles - 1) * n_pairs);
        }
    }
    // ---- self-cleaning: the last workgroup out re-zeroes every polled word ------------------------------------------
    __syncthreads();
    if (tid == 0)
        ctl_s[2] = (__hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) ? 1 : 0;
    __syncthreads();
    if (ctl_s[2] != 0) {
        for (int i = tid; i < (int)(gridDim.x / 2) * 32; i += 256) __hip_atomic_store(pp.ctl + 256 + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid < 8) __hip_atomic_store(class_ticket + tid * 16, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid == 0) __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

constexpr size_t pipe_smem() {
    // stage B is the larger one: 2 in-buffers + h2 + bias + partial sums (stage A: x tile + h0 + 2 biases)
    return (16 + (size_t)3 * HLF + H + 4 * 32 * TR) * sizeof(float);
}

}  // namespace

size_t ape_mlp_pipe_ring_bytes(int n_cus) { return (size_t)(n_cus / 2) * NSLOT * HLF * sizeof(float); }
size_t ape_mlp_pipe_ctl_words(int n_cus) { return 256 + (size_t)(n_cus / 2) * 32; }
bool ape_mlp_pipe_supported(int Hd, int n_hidden, int KXd, int O) { return Hd == H && n_hidden == 2 && KXd == KX && O <= 32; }

hipError_t ape_prepare_mlp_pipe() {
    static_assert(pipe_smem() <= APE_LDS_BYTES, "LDS layout exceeds a CU");
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&ape_mlp_pipe), hipFuncAttributeMaxDynamicSharedMemorySize, APE_LDS_BYTES);
}

// one workgroup per CU, all of them resident at once (pairs hand tiles over through bounded spins): grid = the device's CU count
// rounded down to whole block-index classes of pairs
hipError_t ape_launch_mlp_pipe(const MlpParams& q, const float* wa0, const float* wa1, const float* wb2, const float* wbo, float* ring,
                               size_t ring_bytes, unsigned* ctl, int n_cus, hipStream_t stream) {
    PipeParams pp{};
    static const unsigned diag = getenv("APE_PIPE_DIAG") ? (unsigned)atoi(getenv("APE_PIPE_DIAG")) : 0u;
    pp.diag = diag;
    pp.m = q; pp.wa0 = wa0; pp.wa1 = wa1; pp.wb2 = wb2; pp.wbo = wbo; pp.ring = ring; pp.ring_bytes = ring_bytes; pp.ctl = ctl;
    const int grid = (n_cus / 16) * 16;
    hipLaunchKernelGGL(ape_mlp_pipe, dim3(grid), dim3(256), pipe_smem(), stream, pp);
    return hipGetLastError();
}
