// Microbenchmark: ONE dependent chain of v_mfma_f32_32x32x2_f32 per wave (one wave per SIMD), with a pinned VALU sequence
// between consecutive MFMAs (pass-through asm operands keep the sequence where it is written).  Question: what does VALU work
// in the shadow of a dependent 64-cycle MFMA cost, by kind and amount?   hipcc -O3 --offload-arch=gfx950 chain32_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters) {
    const int lane = threadIdx.x & 63;
    __shared__ __attribute__((aligned(16))) float lds[16 * 256 + 64];
    for (int i = threadIdx.x; i < 16 * 256 + 64; i += 256) lds[i] = 0.001f * (i % 97);
    __syncthreads();
    unsigned sc = 0;
    f32x4v tq = {0.0f, 0.0f, 0.0f, 0.0f};
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    float w = 0.001f * lane, a = 0.002f * lane;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = 0.1f * (lane + i) + 0.01f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 64; ++q) {
            asm volatile("v_mfma_f32_32x32x2_f32 %0, %9, %10, %0"
                         : "+v"(acc), "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) : "v"(w), "v"(a));
            if (MODE == 1) v[0] = v[0] * 1.0001f;
            if (MODE == 2) { v[0] *= 1.0001f; v[1] *= 1.0001f; v[2] *= 1.0001f; v[3] *= 1.0001f; }
            if (MODE == 3) v[0] = __builtin_amdgcn_exp2f(v[0]);
            if (MODE == 4) v[0] = __builtin_amdgcn_exp2f(v[1] * -1.44f);
            if (MODE == 5) { v[0] = __builtin_amdgcn_exp2f(v[1] * -1.44f); v[2] = __builtin_amdgcn_rcpf(1.0f + v[3]); }
            if (MODE == 6) { v[0] *= 1.0001f; v[1] *= 1.0001f; v[2] *= 1.0001f; v[3] *= 1.0001f; v[4] *= 1.0001f; v[5] *= 1.0001f; v[6] *= 1.0001f; v[7] *= 1.0001f; }
            if (MODE == 7) { v[0] = __builtin_amdgcn_exp2f(v[0]); v[1] = __builtin_amdgcn_exp2f(v[1]); v[2] = __builtin_amdgcn_rcpf(v[2]); v[3] = __builtin_amdgcn_rcpf(v[3]); }
            if (MODE == 8) { v[0] = __builtin_amdgcn_exp2f(v[0]); v[1] = __builtin_amdgcn_exp2f(v[1]); }
            if (MODE == 9) { v[0] = fmaf(v[1], v[2], v[0]); v[3] = v[4] + v[5]; }
            if (MODE == 10 && (q & 3) == 3) {       // the product kernel's k-block overhead: one 16-byte LDS read + its wait per 4 MFMAs
                f32x4v t = *reinterpret_cast<const f32x4v*>(lds + ((q >> 2) & 15) * 256 + 4 * lane);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
            }
            if (MODE == 11 && (q & 3) == 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (MODE == 12) asm volatile("s_add_u32 %0, %0, 1\n\ts_and_b32 %0, %0, 0xffff" : "+s"(sc));
            if (MODE == 13 && (q & 3) == 3) {       // the read issued, waited for one block later (double buffer)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                v[0] = tq[0]; v[1] = tq[1]; v[2] = tq[2]; v[3] = tq[3];
                tq = *reinterpret_cast<const f32x4v*>(lds + ((q >> 2) & 15) * 256 + 4 * lane);
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[5] + s + (float)sc + tq[0];
    if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<unsigned long long*>(out + 256 * 256)[0] = t1 - t0;
}

template <int MODE>
void run(float* d, const char* what) {
    const int iters = 200;
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(256), 0, 0, d, iters);
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(256), 0, 0, d, iters);
    hipDeviceSynchronize();
    unsigned long long c = 0;
    hipMemcpy(&c, d + 256 * 256, 8, hipMemcpyDeviceToHost);
    // s_memtime counts at 100 MHz on this part: report in that unit per MFMA and let the caller scale by the mode-0 row (= 64 cycles)
    printf("mode %d  %-44s  %8.3f ticks per MFMA\n", MODE, what, (double)c / (iters * 64.0));
}

int main() {
    float* d;
    hipMalloc(&d, (256 * 256 + 16) * sizeof(float));
    run<0>(d, "bare dependent chain (= 64 shader cycles)");
    run<1>(d, "1 v_mul");
    run<2>(d, "4 independent v_mul");
    run<6>(d, "8 independent v_mul");
    run<9>(d, "v_fma + v_add");
    run<3>(d, "1 v_exp (independent)");
    run<4>(d, "v_mul -> v_exp");
    run<8>(d, "2 independent v_exp");
    run<5>(d, "v_mul -> v_exp, v_add -> v_rcp");
    run<7>(d, "2 v_exp + 2 v_rcp independent");
    run<12>(d, "2 SALU instructions per MFMA");
    run<11>(d, "s_waitcnt lgkmcnt(0) per 4 MFMAs");
    run<10>(d, "ds_read_b128 + wait + 4 v_mov per 4 MFMAs");
    run<13>(d, "the same, read a block ahead");
    hipFree(d);
    return 0;
}
