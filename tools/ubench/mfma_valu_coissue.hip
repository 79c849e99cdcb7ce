// Microbenchmark: one wave per SIMD issuing v_mfma_f32_16x16x4_f32 back to back (4 independent
// accumulators, B operand resident in registers, A operand re-read from LDS per 16 MFMAs), with
// K independent VALU instructions (exp2 / rcp / fma chain) scheduled between consecutive MFMAs.
// Question: how many VALU instructions per MFMA gap are free?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int VPM>   // VALU ops per MFMA
__global__ __launch_bounds__(256, 1) void k(float* out, const float* w_in, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[64 * 264];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 64 * 264; i += 256) lds[i] = 0.001f * (i % 97);
    float w[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) w[i] = w_in[i * 64 + lane];
    __syncthreads();
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = 0.1f * (lane + i);
    const float* src = lds + (lane & 15) * 264 + 4 * (lane >> 4);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            f32x4 a[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) a[mt] = *reinterpret_cast<const f32x4*>(src + mt * 16 * 264 + 16 * q);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt][j], w[4 * q + j], acc[mt], 0, 0, 0);
#pragma unroll
                    for (int e = 0; e < VPM; ++e) {       // independent VALU stream: sigmoid-like chains
                        const int s = (j * 4 + mt + e) & 7;
                        if (e % 3 == 0) v[s] = __builtin_amdgcn_exp2f(v[s] * -1.4426950f);
                        else if (e % 3 == 1) v[s] = __builtin_amdgcn_rcpf(1.0f + v[s]);
                        else v[s] = fmaf(v[s], 0.5f, 0.25f);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);       // 1 MFMA
                    if (VPM > 0) __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);   // VPM VALU
                }
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + s;
    if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<unsigned long long*>(out + 256 * 256)[0] = t1 - t0;
}

// Two waves per SIMD: waves 0-3 (one per SIMD) run the bare MFMA loop, waves 4-7 (the second wave of each SIMD) a
// bare transcendental loop of `vops` exp2/rcp pairs per trip.  Question: does ANOTHER wave's VALU work slow the
// MFMA wave down the way the same wave's does?
__global__ __launch_bounds__(512, 1) void k2(float* out, const float* w_in, int iters, int valu_trips, unsigned mfma_mask) {
    __shared__ __attribute__((aligned(16))) float lds[64 * 264];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 64 * 264; i += 512) lds[i] = 0.001f * (i % 97);
    if (blockIdx.x == 0 && lane == 0) {          // which SIMD did this wave land on (HW_ID bits 5:4)?
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        reinterpret_cast<unsigned*>(out + 256 * 512 + 8)[wave] = (hw >> 4) & 3;
    }
    __syncthreads();
    if ((mfma_mask >> wave) & 1) {
        float w[64];
#pragma unroll
        for (int i = 0; i < 64; ++i) w[i] = w_in[i * 64 + lane];
        f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        const float* src = lds + (lane & 15) * 264 + 4 * (lane >> 4);
        unsigned long long t0 = __builtin_amdgcn_s_memtime();
        // same explicitly double-buffered loop as the product kernel: A fragments of block q+1 fetched before the
        // MFMAs of block q (a plain loop exposes the LDS latency once per block: 42-50 instead of 32 cycles per MFMA)
        f32x4 a_cur[4], a_nxt[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) a_cur[mt] = *reinterpret_cast<const f32x4*>(src + mt * 16 * 264);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) a_nxt[mt] = *reinterpret_cast<const f32x4*>(src + mt * 16 * 264 + 16 * ((q + 1) & 15));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
                        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[mt][j], w[4 * q + j], acc[mt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) a_cur[mt] = a_nxt[mt];
            }
        }
        unsigned long long t1 = __builtin_amdgcn_s_memtime();
        out[blockIdx.x * 512 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
        if (lane == 0 && blockIdx.x == 0 && wave == __builtin_ctz(mfma_mask)) reinterpret_cast<unsigned long long*>(out + 256 * 512)[0] = t1 - t0;
    } else {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = 0.1f * (lane + i);
        unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < valu_trips; ++it) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[e] = __builtin_amdgcn_exp2f(v[e] * -1.4426950f);
                v[e] = __builtin_amdgcn_rcpf(1.0f + v[e]);
            }
        }
        unsigned long long t1 = __builtin_amdgcn_s_memtime();
        float s = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) s += v[i];
        out[blockIdx.x * 512 + threadIdx.x] = s;
        if (lane == 0 && blockIdx.x == 0 && wave == __builtin_ctz(~mfma_mask & 0xFF)) reinterpret_cast<unsigned long long*>(out + 256 * 512)[1] = t1 - t0;
    }
}

void run2(float* out, float* w, int iters, int valu_trips, unsigned mfma_mask = 0x0F) {
    (void)hipMemset(out + 256 * 512, 0, 64);
    hipLaunchKernelGGL(k2, dim3(256), dim3(512), 0, 0, out, w, iters, valu_trips, mfma_mask);
    (void)hipDeviceSynchronize();
    hipLaunchKernelGGL(k2, dim3(256), dim3(512), 0, 0, out, w, iters, valu_trips, mfma_mask);
    (void)hipDeviceSynchronize();
    unsigned long long cyc[2]; (void)hipMemcpy(cyc, out + 256 * 512, 16, hipMemcpyDeviceToHost);
    unsigned simd[8]; (void)hipMemcpy(simd, out + 256 * 512 + 8, 32, hipMemcpyDeviceToHost);
    printf("MFMA waves mask 0x%02x, SIMD of waves 0..7: %u%u%u%u%u%u%u%u | ", mfma_mask, simd[0], simd[1], simd[2], simd[3], simd[4],
           simd[5], simd[6], simd[7]);
    printf("2 waves/SIMD, helper wave runs %d x 16 transcendentals: MFMA wave %.1f cycles per MFMA (%llu cycles), "
           "helper wave %llu cycles = %.1f per transcendental\n", valu_trips, cyc[0] / (256.0 * iters), cyc[0], cyc[1],
           valu_trips ? cyc[1] / (16.0 * valu_trips) : 0.0);
}

// Symmetric split: every wave alternates a burst of MFMAs with a burst of gate-math-like VALU work (exp2 / rcp / fma).
// WAVES = 4: one wave per SIMD does 256 MFMAs + 160 VALU per trip (the one-role kernel's shape: serial).
// WAVES = 8: two waves per SIMD do 128 MFMAs + 80 VALU each per trip, the second wave of a SIMD starting with its VALU
// burst (phase offset) -- does the pair keep the matrix pipe busier than one wave doing both jobs back to back?
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64, 1) void k3(float* out, const float* w_in, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[64 * 264];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 64 * 264; i += WAVES * 64) lds[i] = 0.001f * (i % 97);
    constexpr int NB = (WAVES == 4) ? 16 : 8;            // k-blocks of 16 MFMAs per trip
    constexpr int NV = (WAVES == 4) ? 160 : 80;          // VALU instructions per trip (half of them transcendental)
    float w[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) w[i] = w_in[i * 64 + lane];
    __syncthreads();
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = 0.1f * (lane + i);
    const float* src = lds + (lane & 15) * 264 + 4 * (lane >> 4);
    auto valu_burst = [&]() {
#pragma unroll
        for (int e = 0; e < NV / 4; ++e) {
            const int s = e & 7;
            v[s] = __builtin_amdgcn_exp2f(v[s] * -1.4426950f);
            v[s] = __builtin_amdgcn_rcpf(1.0f + v[s]);
        }
    };
    auto mfma_burst = [&]() {
        f32x4 a_cur[4], a_nxt[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) a_cur[mt] = *reinterpret_cast<const f32x4*>(src + mt * 16 * 264);
#pragma unroll
        for (int q = 0; q < NB; ++q) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) a_nxt[mt] = *reinterpret_cast<const f32x4*>(src + mt * 16 * 264 + 16 * ((q + 1) & 15));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[mt][j], w[(4 * q + j) & 31], acc[mt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) a_cur[mt] = a_nxt[mt];
        }
    };
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (WAVES == 8 && wave >= 4) valu_burst();           // phase offset of the second wave of each SIMD
    for (int it = 0; it < iters; ++it) {
        mfma_burst();
        __builtin_amdgcn_sched_barrier(0);
        valu_burst();
        __builtin_amdgcn_sched_barrier(0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sres = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) sres += v[i];
    out[blockIdx.x * WAVES * 64 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + sres;
    if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<unsigned long long*>(out + 256 * 512)[0] = t1 - t0;
}

template <int WAVES>
void run3(float* out, float* w, int iters) {
    hipLaunchKernelGGL(k3<WAVES>, dim3(256), dim3(WAVES * 64), 0, 0, out, w, iters);
    (void)hipDeviceSynchronize();
    hipEvent_t ea, eb; (void)hipEventCreate(&ea); (void)hipEventCreate(&eb);
    (void)hipEventRecord(ea);
    hipLaunchKernelGGL(k3<WAVES>, dim3(256), dim3(WAVES * 64), 0, 0, out, w, iters);
    (void)hipEventRecord(eb); (void)hipEventSynchronize(eb);
    float ms = 0; (void)hipEventElapsedTime(&ms, ea, eb);
    unsigned long long cyc; (void)hipMemcpy(&cyc, out + 256 * 512, 8, hipMemcpyDeviceToHost);
    printf("   whole kernel %.3f ms = %.1f TFLOP/s of MFMA work on the chip; ", ms, 256.0 * 4 * 256 * iters * 2048 / (ms * 1e-3) / 1e12);
    // per SIMD and trip: 256 MFMAs (= 8192 pipe cycles) + 160 VALU instructions, whatever the number of waves
    printf("%d wave(s) per SIMD, MFMA bursts alternating with VALU bursts: %.0f cycles per trip of 256 MFMAs + 160 VALU (pipe alone: 8192)\n",
           WAVES / 4, (double)cyc / iters);
}

// Symmetric split: every wave alternates a burst of MFMAs with a burst of gate-math-like VALU work (exp2 / rcp / fma).
// WAVES = 4: one wave per SIMD does 256 MFMAs + 160 VALU per trip (the one-role kernel's shape: serial).
// WAVES = 8: two waves per SIMD do 128 MFMAs + 80 VALU each per trip, the second wave of a SIMD starting with its VALU
// burst (phase offset) -- does the pair keep the matrix pipe busier than one wave doing both jobs back to back?
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64, 1) void k3t(float* out, const float* w_in, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[64 * 264];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 64 * 264; i += WAVES * 64) lds[i] = 0.001f * (i % 97);
    constexpr int NB = (WAVES == 4) ? 16 : 8;            // k-blocks of 16 MFMAs per trip
    constexpr int NV = (WAVES == 4) ? 160 : 80;          // VALU instructions per trip (half of them transcendental)
    float w[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) w[i] = w_in[i * 64 + lane];
    __syncthreads();
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = 0.1f * (lane + i);
    const float* src = lds + (lane & 15) * 264 + 4 * (lane >> 4);
    auto valu_burst = [&]() {
        __builtin_amdgcn_s_setprio(3);
#pragma unroll
        for (int e = 0; e < NV / 4; ++e) {
            const int s = e & 7;
            v[s] = __builtin_amdgcn_exp2f(v[s] * -1.4426950f);
            v[s] = __builtin_amdgcn_rcpf(1.0f + v[s]);
        }
        __builtin_amdgcn_s_setprio(0);
    };
    auto mfma_burst = [&]() {
        f32x4 a_cur[4], a_nxt[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) a_cur[mt] = *reinterpret_cast<const f32x4*>(src + mt * 16 * 264);
#pragma unroll
        for (int q = 0; q < NB; ++q) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) a_nxt[mt] = *reinterpret_cast<const f32x4*>(src + mt * 16 * 264 + 16 * ((q + 1) & 15));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[mt][j], w[(4 * q + j) & 31], acc[mt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) a_cur[mt] = a_nxt[mt];
        }
    };
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long* tl = reinterpret_cast<unsigned long long*>(out + 256 * 512 + 16) + (wave >= 4 ? 64 : 0);
    const bool rec = blockIdx.x == 0 && (wave == 0 || wave == 4) && lane == 0;
    for (int it = 0; it < iters; ++it) {
        if (WAVES == 8) __syncthreads();                     // the two waves of a SIMD start every trip together
        unsigned long long ta, tb, tc;
        if (WAVES == 8 && wave >= 4) {                       // second wave of the SIMD: gate math first, matrix work second
            valu_burst();
            __builtin_amdgcn_sched_barrier(0);
            ta = __builtin_amdgcn_s_memtime();
            mfma_burst();
            __builtin_amdgcn_sched_barrier(0);
            tb = tc = __builtin_amdgcn_s_memtime();
        } else {
            ta = __builtin_amdgcn_s_memtime();
            mfma_burst();
            __builtin_amdgcn_sched_barrier(0);
            tb = __builtin_amdgcn_s_memtime();
            valu_burst();
            __builtin_amdgcn_sched_barrier(0);
            tc = __builtin_amdgcn_s_memtime();
        }
        if (rec && it >= 100 && it < 110) { tl[(it - 100) * 3] = ta - t0; tl[(it - 100) * 3 + 1] = tb - t0; tl[(it - 100) * 3 + 2] = tc - t0; }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sres = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) sres += v[i];
    out[blockIdx.x * WAVES * 64 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + sres;
    if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<unsigned long long*>(out + 256 * 512)[0] = t1 - t0;
}

template <int WAVES>
void run3t(float* out, float* w, int iters) {
    hipLaunchKernelGGL(k3t<WAVES>, dim3(256), dim3(WAVES * 64), 0, 0, out, w, iters);
    (void)hipDeviceSynchronize();
    hipEvent_t ea, eb; (void)hipEventCreate(&ea); (void)hipEventCreate(&eb);
    (void)hipEventRecord(ea);
    hipLaunchKernelGGL(k3t<WAVES>, dim3(256), dim3(WAVES * 64), 0, 0, out, w, iters);
    (void)hipEventRecord(eb); (void)hipEventSynchronize(eb);
    float ms = 0; (void)hipEventElapsedTime(&ms, ea, eb);
    unsigned long long tl[128]; (void)hipMemcpy(tl, out + 256 * 512 + 16, sizeof(tl), hipMemcpyDeviceToHost);
    for (int wv = 0; wv < (WAVES == 8 ? 2 : 1); ++wv) {
        printf("   timeline wave %d (mfma start, mfma len, valu len):", wv * 4);
        for (int i = 0; i < 6; ++i) printf("  %llu %llu %llu |", tl[wv * 64 + i * 3] - tl[0], tl[wv * 64 + i * 3 + 1] - tl[wv * 64 + i * 3], tl[wv * 64 + i * 3 + 2] - tl[wv * 64 + i * 3 + 1]);
        printf("\n");
    }
    unsigned long long cyc; (void)hipMemcpy(&cyc, out + 256 * 512, 8, hipMemcpyDeviceToHost);
    printf("   whole kernel %.3f ms = %.1f TFLOP/s of MFMA work on the chip; ", ms, 256.0 * 4 * 256 * iters * 2048 / (ms * 1e-3) / 1e12);
    // per SIMD and trip: 256 MFMAs (= 8192 pipe cycles) + 160 VALU instructions, whatever the number of waves
    printf("%d wave(s) per SIMD, MFMA bursts alternating with VALU bursts: %.0f cycles per trip of 256 MFMAs + 160 VALU (pipe alone: 8192)\n",
           WAVES / 4, (double)cyc / iters);
}

template <int VPM>
void run(float* out, float* w, int iters) {
    hipLaunchKernelGGL(k<VPM>, dim3(256), dim3(256), 0, 0, out, w, iters);
    hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<VPM>, dim3(256), dim3(256), 0, 0, out, w, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    unsigned long long cyc; hipMemcpy(&cyc, out + 256 * 256, 8, hipMemcpyDeviceToHost);
    const double mfmas = 256.0 * iters;
    printf("VALU/MFMA=%d : %.1f cycles per MFMA (in-kernel), %.3f ms, %.1f TFLOP/s chip\n", VPM, cyc / mfmas, ms,
           256.0 * 4 * mfmas * 2048 / (ms * 1e-3) / 1e12);
}

int main() {
    float *out, *w;
    hipMalloc(&out, (256 * 512 + 1024) * 4); hipMalloc(&w, 64 * 64 * 4);
    std::vector<float> hw(64 * 64); for (size_t i = 0; i < hw.size(); ++i) hw[i] = 0.01f * (i % 31) - 0.1f;
    hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    run<0>(out, w, 200); run<1>(out, w, 200); run<2>(out, w, 200); run<3>(out, w, 200); run<4>(out, w, 200);
    run<6>(out, w, 200); run<8>(out, w, 200);
    // 200 iterations x 256 MFMAs x 32 cycles = 1.64M cycles of MFMA work per wave
    run3<4>(out, w, 200); run3<8>(out, w, 200);
    run3t<4>(out, w, 200); run3t<8>(out, w, 200);
    run2(out, w, 200, 0); run2(out, w, 200, 3000); run2(out, w, 200, 12000);
    run2(out, w, 200, 0, 0x55); run2(out, w, 200, 3000, 0x55);      // MFMA on waves 0,2,4,6
    run2(out, w, 200, 0, 0x01); run2(out, w, 200, 3000, 0x01);      // ONE MFMA wave, seven helpers
    run2(out, w, 200, 0, 0x33); run2(out, w, 200, 3000, 0x33);
    return 0;
}
