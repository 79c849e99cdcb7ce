// How many independent accumulator chains does a wave need to keep the matrix core of its SIMD busy?  One wave per SIMD (4 per
// workgroup, 1 workgroup per CU), N MFMAs round-robin over C chains, cycles per MFMA from s_memtime.
//   hipcc -O3 --offload-arch=gfx950 tools/experiments/mfma_chain_rate.hip -o /tmp/mfma_chain_rate && /tmp/mfma_chain_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int C>
__global__ __launch_bounds__(256, 1) void k16(unsigned long long* out, float a, float b, int iters) {
    f32x4 acc[C];
    for (int c = 0; c < C; ++c) acc[c] = f32x4{0, 0, 0, 0};
    float wa = a + threadIdx.x, wb = b;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 32 / C; ++r)
#pragma unroll
            for (int c = 0; c < C; ++c) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[c]) : "v"(wa), "v"(wb));
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int c = 0; c < C; ++c) s += acc[c][0];
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = (unsigned long long)s; }
}
template <int C>
__global__ __launch_bounds__(256, 1) void k32(unsigned long long* out, float a, float b, int iters) {
    f32x16 acc[C];
    for (int c = 0; c < C; ++c) for (int i = 0; i < 16; ++i) acc[c][i] = 0;
    float wa = a + threadIdx.x, wb = b;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 32 / C; ++r)
#pragma unroll
            for (int c = 0; c < C; ++c) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[c]) : "v"(wa), "v"(wb));
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int c = 0; c < C; ++c) s += acc[c][0];
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = (unsigned long long)s; }
}
// two chains, and between every two MFMAs V plain VALU instructions (v_fma_f32) and X transcendentals (v_exp_f32) on other registers
template <int V, int X>
__global__ __launch_bounds__(256, 1) void k16v(unsigned long long* out, float a, float b, int iters) {
    f32x4 acc[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
    float wa = a + threadIdx.x, wb = b, v0 = a, v1 = b;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 32; ++r) {
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[r & 1]) : "v"(wa), "v"(wb));
#pragma unroll
            for (int i = 0; i < V; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v0) : "v"(wb));
#pragma unroll
            for (int i = 0; i < X; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(v1));
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = (unsigned long long)(acc[0][0] + acc[1][0] + v0 + v1); }
}
// the steady-state block of lstm_cluster32.hip: four 32x32x2 MFMAs of one chain, MODE 0: nothing else; 1: an s_nop 0 behind each of the
// first three (what hipcc puts between inline-asm MFMAs); 2: + one ds_read_b128 and a counted wait per block; 3: the ds_read and the wait
// only; 4: all four MFMAs in ONE asm statement + the ds_read and the wait
template <int MODE>
__global__ __launch_bounds__(256, 1) void k32blk(unsigned long long* out, float a, float b, int iters) {
    __shared__ f32x4 lds[1024];
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0;
    lds[threadIdx.x] = f32x4{a, b, a, b};
    __syncthreads();
    float wa = a + threadIdx.x;
    f32x4 f0 = lds[threadIdx.x], f1 = f0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            // MODE 5: lstm_cluster32.hip's fragment addresses ([window 32][8 units]: lane (n, hh) reads 16 bytes at n * 32 + hh * 16);
            // MODE 6: [half][window][4 units]: n * 16 + hh * 512
            const unsigned lane_ = threadIdx.x & 63, n_ = lane_ & 31, hh_ = lane_ >> 5;
            const unsigned a_ = MODE == 5 ? n_ * 32 + hh_ * 16 : MODE == 6 ? n_ * 16 + hh_ * 512 : (threadIdx.x & 63) * 16;
            if (MODE >= 2) asm volatile("ds_read_b128 %0, %1" : "=v"(f1) : "v"((unsigned)(a_ + (MODE >= 5 ? 0u : (threadIdx.x >> 6) * 1024u) + (r & 3) * 4096)));
            if (MODE == 4) {
                asm volatile("s_waitcnt lgkmcnt(1)\n\tv_mfma_f32_32x32x2_f32 %0, %1, %2, %0\n\tv_mfma_f32_32x32x2_f32 %0, %1, %3, %0\n\t"
                             "v_mfma_f32_32x32x2_f32 %0, %1, %4, %0\n\tv_mfma_f32_32x32x2_f32 %0, %1, %5, %0"
                             : "+v"(acc) : "v"(wa), "v"(f0[0]), "v"(f0[1]), "v"(f0[2]), "v"(f0[3]));
            } else {
                if (MODE >= 2) asm volatile("s_waitcnt lgkmcnt(1)");
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(wa), "v"(f0[j]));
                    if ((MODE == 1 || MODE == 2) && j < 3) asm volatile("s_nop 0");
                }
            }
            if (MODE >= 2) { asm volatile("" : "+v"(f1)); f0 = f1; }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = (unsigned long long)(acc[0] + f0[0]); }
}
// the same with TWO waves per SIMD (512 threads): cycles per MFMA of ONE wave; 64 = the matrix core never idles (2 x 32)
template <int V, int X>
__global__ __launch_bounds__(512, 1) void k16v2(unsigned long long* out, float a, float b, int iters) {
    f32x4 acc[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
    float wa = a + threadIdx.x, wb = b, v0 = a, v1 = b;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 32; ++r) {
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[r & 1]) : "v"(wa), "v"(wb));
#pragma unroll
            for (int i = 0; i < V; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v0) : "v"(wb));
#pragma unroll
            for (int i = 0; i < X; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(v1));
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = (unsigned long long)(acc[0][0] + acc[1][0] + v0 + v1); }
}
// weights as the MFMA's A operand from the accumulation file ("a") against the architectural one ("v"): 32 DIFFERENT registers
template <bool AG>
__global__ __launch_bounds__(256, 1) void k32src(unsigned long long* out, float a, float b, int iters) {
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0;
    float w[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) { w[i] = a + i + threadIdx.x; asm volatile("" : "+v"(w[i])); }
    float wb = b;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 32; ++r) {
            if constexpr (AG) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "a"(w[r]), "v"(wb));
            else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(w[r]), "v"(wb));
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = (unsigned long long)acc[0]; }
}
// four 32x32x2 MFMAs + ONE LDS-DMA instruction (s_mov m0, buffer_load_dwordx4 ... lds: 1 KB global -> LDS) per block: what does the issue
// of a gather piece cost inside an MFMA stream?  MODE 0: MFMAs only; 1: + the DMA; 2: + a global_load_dword (the flag look)
typedef unsigned u32x4_ __attribute__((__vector_size__(16)));
template <int MODE>
__global__ __launch_bounds__(256, 1) void k32dma(unsigned long long* out, float a, float b, int iters, const float* src) {
    __shared__ float lds[8192];
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0;
    float wa = a + threadIdx.x, wb = b;
    const unsigned long long sa = reinterpret_cast<unsigned long long>(src);
    u32x4_ desc;
    desc[0] = __builtin_amdgcn_readfirstlane((unsigned)sa); desc[1] = __builtin_amdgcn_readfirstlane((unsigned)(sa >> 32) & 0xFFFFu);
    desc[2] = 1u << 20; desc[3] = 0x00020000u;
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)reinterpret_cast<unsigned long long>(lds) + (threadIdx.x >> 6) * 8192);
    const unsigned soff0 = __builtin_amdgcn_readfirstlane(blockIdx.x * 8192u);
    const unsigned voff = (threadIdx.x & 63) * 16;
    unsigned peek = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(wa), "v"(wb));
            if (MODE == 1)
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen sc1 lds"
                             :: "s"(lds_base + (r & 7) * 1024), "v"(voff), "s"(desc), "s"(soff0 + (unsigned)(r * 1024)) : "memory");
            if (MODE == 2) asm volatile("global_load_dword %0, %1, off sc1" : "=v"(peek) : "v"(src + (threadIdx.x & 31)) : "memory");
        }
        if (MODE != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = (unsigned long long)(acc[0] + lds[threadIdx.x] + peek); }
}
template <typename K>
void run3(const char* name, K kern) {
    unsigned long long* d; (void)hipMalloc(&d, 16);
    float* src; (void)hipMalloc(&src, 4 << 20);
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, d, 1.0f, 0.5f, iters, src);
    unsigned long long h[2]; (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("%s: %.2f cycles per MFMA\n", name, (double)h[0] / (iters * 32.0));
    (void)hipFree(d); (void)hipFree(src);
}
template <typename K>
void run2(const char* name, K kern) {
    unsigned long long* d; (void)hipMalloc(&d, 16);
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, d, 1.0f, 0.5f, iters);
    unsigned long long h[2]; (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, d, 1.0f, 0.5f, iters);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("two waves per SIMD, %s: %.2f cycles per MFMA of one wave; wall clock %.3f ms = %.1f TFLOP/s over 256 x 8 waves\n", name,
           (double)h[0] / (iters * 32.0), ms, 256.0 * 8 * iters * 32.0 * 2048.0 / (ms * 1e-3) / 1e12);
    (void)hipFree(d);
}
template <typename K>
void run(const char* name, K kern, int chains) {
    unsigned long long* d; (void)hipMalloc(&d, 16);
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, d, 1.0f, 0.5f, iters);
    unsigned long long h[2]; (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, d, 1.0f, 0.5f, iters);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%s, %d chain(s): %.2f cycles per MFMA; wall clock %.3f ms for %d MFMAs per wave\n", name, chains, (double)h[0] / (iters * 32.0), ms, iters * 32);
    (void)hipFree(d);
}
int main() {
    run("v_mfma_f32_16x16x4_f32", k16<1>, 1); run("v_mfma_f32_16x16x4_f32", k16<2>, 2); run("v_mfma_f32_16x16x4_f32", k16<4>, 4);
    run("v_mfma_f32_32x32x2_f32", k32<1>, 1); run("v_mfma_f32_32x32x2_f32", k32<2>, 2); run("v_mfma_f32_32x32x2_f32", k32<4>, 4);
    run("16x16x4 x 2 chains + 1 v_fma between", k16v<1, 0>, 2); run("16x16x4 x 2 chains + 4 v_fma between", k16v<4, 0>, 2);
    run("16x16x4 x 2 chains + 6 v_fma between", k16v<6, 0>, 2); run("16x16x4 x 2 chains + 1 v_exp between", k16v<0, 1>, 2);
    run("16x16x4 x 2 chains + 2 v_exp between", k16v<0, 2>, 2); run("16x16x4 x 2 chains + 1 v_exp + 2 v_fma between", k16v<2, 1>, 2);
    run("32x32x2 block of 4, bare", k32blk<0>, 1); run("32x32x2 block of 4 + s_nop 0 x 3", k32blk<1>, 1);
    run("32x32x2 block of 4 + s_nop 0 x 3 + ds_read_b128 + wait", k32blk<2>, 1); run("32x32x2 block of 4 + ds_read_b128 + wait", k32blk<3>, 1);
    run("32x32x2 block of 4 in one asm + ds_read_b128 + wait", k32blk<4>, 1);
    run("32x32x2 block of 4 + ds_read_b128 at the kernel's addresses [window][8]", k32blk<5>, 1);
    run("32x32x2 block of 4 + ds_read_b128 at [half][window][4]", k32blk<6>, 1);
    run("32x32x2, 32 different weights from VGPRs", k32src<false>, 1); run("32x32x2, 32 different weights from AGPRs", k32src<true>, 1);
    run3("32x32x2 block of 4 (DMA test), MFMAs only", k32dma<0>); run3("32x32x2 block of 4 + one LDS-DMA piece (1 KB) per block", k32dma<1>);
    run3("32x32x2 block of 4 + one global_load_dword sc1 per block", k32dma<2>);
    run2("bare", k16v2<0, 0>); run2("1 v_fma between", k16v2<1, 0>); run2("4 v_fma between", k16v2<4, 0>);
    run2("1 v_exp between", k16v2<0, 1>); run2("2 v_exp between", k16v2<0, 2>); run2("1 v_exp + 2 v_fma between", k16v2<2, 1>);
    return 0;
}
