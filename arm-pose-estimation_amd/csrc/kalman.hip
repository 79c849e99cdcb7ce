// Ensemble Kalman estimator on the GPU: one forward of KalmanSmartwatchModel (reference estimate/kalman_models.py:175-220)
// for S streams, and format_state (:164-173).  SURVEY.md section 8 row f4 (tail).  PARITY UNPINNED: the reference module
// needs bayesian_torch (not installed) and its checkpoint is absent; oracle/kalman_oracle.py restates it, and that is what
// tests/test_kalman.py compares this file with.
//
// Layers (reference names; kalman_models.py:32-34, 70-71, 112-115), row = (stream s, ensemble member e), R = S * E rows:
//   process model   [R, 14 W] -flipout-> 256 -leaky-> -flipout-> 512 -leaky-> -linear-> 14      (:37-50)
//   sensor model    [S, 22 W] -linear-> 256 -leaky-> (every member of a stream reads its stream's row)
//                             -flipout-> 256 -leaky-> -flipout-> 64 -leaky-> -flipout-> 14       (:117-136)
//   per stream: ensemble means, observation noise 14 -> 32 -relu-> 14, (x + 1e-3)^2 + 0.038729833 (:73-80), the 14 x 14
//   innovation, its inverse, the gain and the corrected ensemble (:181-208)
// LinearFlipout (bayesian-torch, as called at :44,46,124,126,128):  y = x mu_W^T + mu_b + ((x * s_in) dW^T + db) * s_out  with
// dW = log1p(exp(rho_W)) * eps_W, eps ~ N(0,1) once per forward (shared by all rows), s_in / s_out = +-1 per element.
//
// Kernels.  kf_perturb: dW, db of the five flipout layers in one launch (Philox + Box-Muller, or injected draws).
// kf_linear<FLIP>: a workgroup owns 16 rows x 64 outputs; the rows (and rows * s_in) are staged in LDS once, every wave
// streams the weights of its 16 outputs from L2 as 16-byte fragments and feeds v_mfma_f32_16x16x4_f32 -- for a flipout
// layer two accumulators (mean and perturbation) over the same A-fragments' two variants.  kf_update: one wave per stream,
// everything in LDS, Gauss-Jordan with partial pivoting for the inverse.  All float32, like the reference's tensors.
// This is latency-sized work (S = 1: 32-48 rows, ~32 MFLOP, 2.3 MB of weights per frame); the kernels are bound by launch
// count and L2 latency, not by any roofline.
#include <new>
#include <vector>

#include "../../include/ape_hip.h"
#include "ape_internal.h"

namespace {

typedef float f32x4v __attribute__((ext_vector_type(4)));

constexpr int DX = 14, RAW = 22;
constexpr int NLAYERS = 9;
// layer ids in blob order
enum { P_B1 = 0, P_B3, P_M2, S_FC2, S_FC3, S_FC5, S_FC6, O_FC1, O_FC2 };
constexpr bool IS_FLIP[NLAYERS] = {true, true, false, false, true, true, true, false, false};

struct KfPerturbParams {
    const float* rho[10];       // segment 2j = weights, 2j + 1 = bias of flipout layer j (in blob order)
    float* delta[10];
    const float* eps[10];       // injected standard-normal draws, or nullptr: Philox
    unsigned start[11];         // prefix sums of the segment lengths
    unsigned long long seed;
};

__device__ __forceinline__ float u01(uint32_t x) { return ((float)(x >> 8) + 1.0f) * 5.9604644775390625e-08f; }   // (0, 1], 24 bits

// standard normal number `idx` of stream (tag, seed): Box-Muller on Philox words
__device__ __forceinline__ float philox_normal(unsigned idx, unsigned tag, unsigned long long seed) {
    uint32_t w[4];
    philox4x32(idx >> 2, tag, 0x4B414C4Du, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), w);
    const int pair = (idx >> 1) & 1;
    const float r = sqrtf(-2.0f * logf(u01(w[2 * pair])));
    const float a = 6.283185307179586f * (float)w[2 * pair + 1] * 2.3283064365386963e-10f;
    return (idx & 1) ? r * sinf(a) : r * cosf(a);
}

__global__ void kf_perturb_kernel(const KfPerturbParams p) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.start[10]) return;
    int seg = 0;
#pragma unroll
    for (int s = 1; s < 10; ++s) seg += (i >= p.start[s]) ? 1 : 0;
    const unsigned j = i - p.start[seg];
    const float e = p.eps[seg] ? p.eps[seg][j] : philox_normal(j, 0x100u + seg, p.seed);
    p.delta[seg][j] = log1pf(expf(p.rho[seg][j])) * e;
}

struct KfLinearParams {
    const float* X;        // [R / x_group, K] row-major
    int x_group;           // row r reads X row r / x_group (the sensor model's repeat over the ensemble)
    int R, K, N;
    const float* Wmu;      // [N, K]
    const float* bmu;      // [N]
    const float* Wd;       // [N, K] perturbation (flipout) or nullptr
    const float* bd;
    const float* sign_in;  // injected [R, K] of +-1, or nullptr: Philox bits
    const float* sign_out; // injected [R, N], or nullptr
    unsigned long long seed;
    int layer;
    int act;               // 0 none, 1 leaky_relu(0.01)
    float* Y;              // [R, N]
};

// +-1 number (r, c) of sign stream `tag`
__device__ __forceinline__ uint32_t sign_word(int r, int c, unsigned tag, unsigned long long seed) {
    uint32_t w[4];
    philox4x32((uint32_t)(c >> 7), (uint32_t)r, tag, 0x5349474Eu, (uint32_t)seed, (uint32_t)(seed >> 32), w);
    return w[(c >> 5) & 3];
}
__device__ __forceinline__ float sign_of(uint32_t word, int c) { return ((word >> (c & 31)) & 1u) ? -1.0f : 1.0f; }

template <bool FLIP>
__global__ __launch_bounds__(256) void kf_linear_kernel(const KfLinearParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Kp = (p.K + 15) & ~15, SX = Kp + 4;
    float* xs = lds;                       // [16][SX]
    float* xf = lds + 16 * SX;             // [16][SX]  x * s_in (flipout)
    const int r0 = blockIdx.y * 16;
    // ---- stage the 16 rows (zero beyond R and K); 4 floats per thread and step
    for (int idx = tid; idx < 16 * (Kp / 4); idx += 256) {
        const int rr = idx / (Kp / 4), k = 4 * (idx - rr * (Kp / 4));
        const int r = r0 + rr;
        f32x4v v = {0.0f, 0.0f, 0.0f, 0.0f};
        if (r < p.R && k < p.K) v = *reinterpret_cast<const f32x4v*>(p.X + (size_t)(r / p.x_group) * p.K + k);
        *reinterpret_cast<f32x4v*>(xs + rr * SX + k) = v;
        if constexpr (FLIP) {
            f32x4v s = {1.0f, 1.0f, 1.0f, 1.0f};
            if (r < p.R && k < p.K) {
                if (p.sign_in) s = *reinterpret_cast<const f32x4v*>(p.sign_in + (size_t)r * p.K + k);
                else {
                    const uint32_t w = sign_word(r, k, 0x200u + 2 * p.layer, p.seed);
#pragma unroll
                    for (int j = 0; j < 4; ++j) s[j] = sign_of(w, k + j);
                }
            }
            *reinterpret_cast<f32x4v*>(xf + rr * SX + k) = v * s;
        }
    }
    __syncthreads();
    // ---- this wave: outputs n0 .. n0+15; lane (col = lane & 15, k-group g = lane >> 4)
    const int n0 = blockIdx.x * 64 + wave * 16;
    if (n0 >= p.N) return;
    const int col = lane & 15, g = lane >> 4;
    const int n = n0 + col;
    const bool n_ok = n < p.N;
    f32x4v acc = {0.0f, 0.0f, 0.0f, 0.0f}, accd = {0.0f, 0.0f, 0.0f, 0.0f};
    const float* wmu = p.Wmu + (size_t)(n_ok ? n : 0) * p.K;
    const float* wd = FLIP ? p.Wd + (size_t)(n_ok ? n : 0) * p.K : nullptr;
    // four k-blocks per trip, their weight fragments requested together: one dependent L2 round trip per trip instead of per block
    // (these layers are 3-24 workgroups: latency, not bandwidth)
    constexpr int UQ = 4;
    for (int q0 = 0; q0 < Kp / 16; q0 += UQ) {
        f32x4v b[UQ], bdv[UQ];
#pragma unroll
        for (int u = 0; u < UQ; ++u) {
            const int k = 16 * (q0 + u) + 4 * g;
            const bool k_ok = n_ok && q0 + u < Kp / 16 && k < p.K;
            b[u] = f32x4v{0.0f, 0.0f, 0.0f, 0.0f};
            bdv[u] = b[u];
            if (k_ok) b[u] = *reinterpret_cast<const f32x4v*>(wmu + k);
            if constexpr (FLIP) {
                if (k_ok) bdv[u] = *reinterpret_cast<const f32x4v*>(wd + k);
            }
        }
#pragma unroll
        for (int u = 0; u < UQ; ++u) {
            if (q0 + u >= Kp / 16) break;                            // uniform
            const int k = 16 * (q0 + u) + 4 * g;
            const f32x4v a = *reinterpret_cast<const f32x4v*>(xs + col * SX + k);
            f32x4v af = a;
            if constexpr (FLIP) af = *reinterpret_cast<const f32x4v*>(xf + col * SX + k);
            // MFMA j of the block multiplies k = 16 q + 4 g + j: A lane (row = lane & 15, g) and B lane (col = lane & 15, g) agree
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[u][j], acc, 0, 0, 0);
                if constexpr (FLIP) accd = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j], bdv[u][j], accd, 0, 0, 0);
            }
        }
    }
    // ---- epilogue: acc[i] = C[row 4 g + i][col]
    if (!n_ok) return;
    const float bm = p.bmu[n];
    const float bdl = FLIP ? p.bd[n] : 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + 4 * g + i;
        if (r >= p.R) continue;
        float y = acc[i] + bm;
        if constexpr (FLIP) {
            const float so = p.sign_out ? p.sign_out[(size_t)r * p.N + n] : sign_of(sign_word(r, n, 0x201u + 2 * p.layer, p.seed), n);
            y += (accd[i] + bdl) * so;
        }
        if (p.act == 1) y = y >= 0.0f ? y : 0.01f * y;
        p.Y[(size_t)r * p.N + n] = y;
    }
}

struct KfUpdateParams {
    const float* pred;     // [S, E, 14] process model output
    const float* ensz;     // [S, E, 14] sensor model output
    const float* w1; const float* b1;   // observation noise fc1 [32,14], [32]
    const float* w2; const float* b2;   // fc2 [14,32], [14]
    float* corrected;      // [S, E, 14]
    float* m_corrected;    // [S, 14]
    float* m_pred;         // [S, 14]
    float* z;              // [S, 14]
    float* ensz_out;       // [S, E, 14] or nullptr (already where the caller wants it)
    int S, E;
    int* singular;         // set to 1 when a pivot is exactly zero
};

constexpr int MAXE = 128;

__global__ __launch_bounds__(64) void kf_update_kernel(const KfUpdateParams p) {
    __shared__ float sp[MAXE * DX], sz[MAXE * DX];       // state_pred -> A (anomalies); ensemble_z -> (y - H X)
    __shared__ float mp[DX], mz[DX], rd[DX], hid[32];
    __shared__ float P[DX * DX], M[DX * 2 * DX], Kg[DX * DX];
    __shared__ int piv;
    const int s = blockIdx.x, tid = threadIdx.x, E = p.E;
    const float* pred = p.pred + (size_t)s * E * DX;
    const float* ez = p.ensz + (size_t)s * E * DX;
    for (int i = tid; i < E * DX; i += 64) { sp[i] = pred[i]; sz[i] = ez[i]; }
    __syncthreads();
    if (tid < 2 * DX) {                                 // ensemble means (kalman_models.py:183, :135)
        const float* src = tid < DX ? sp : sz;
        const int c = tid < DX ? tid : tid - DX;
        float a = 0.0f;
        for (int e = 0; e < E; ++e) a += src[e * DX + c];
        a /= (float)E;
        if (tid < DX) mp[c] = a; else mz[c] = a;
    }
    __syncthreads();
    if (tid < 32) {                                     // observation noise (:73-80)
        float a = p.b1[tid];
        for (int c = 0; c < DX; ++c) a += p.w1[tid * DX + c] * mz[c];
        hid[tid] = fmaxf(a, 0.0f);
    }
    __syncthreads();
    if (tid < DX) {
        float a = p.b2[tid];
        for (int c = 0; c < 32; ++c) a += p.w2[tid * 32 + c] * hid[c];
        a += 1e-3f;
        rd[tid] = a * a + 0.038729833f;
        p.m_pred[(size_t)s * DX + tid] = mp[tid];
        p.z[(size_t)s * DX + tid] = mz[tid];
    }
    // D = y - H X (per member), then A = X - mean in place (:184, :206)
    for (int i = tid; i < E * DX; i += 64) {
        const float x = sp[i];
        sz[i] = sz[i] - x;
        sp[i] = x - mp[i % DX];
    }
    __syncthreads();
    const float inv_nm1 = 1.0f / (float)(E - 1);
    for (int idx = tid; idx < DX * DX; idx += 64) {     // P = A^T A / (E - 1)   (:200, :203)
        const int i = idx / DX, j = idx - i * DX;
        float a = 0.0f;
        for (int e = 0; e < E; ++e) a += sp[e * DX + i] * sp[e * DX + j];
        a *= inv_nm1;
        P[idx] = a;
        M[i * 2 * DX + j] = a + (i == j ? rd[i] : 0.0f);        // innovation | identity
        M[i * 2 * DX + DX + j] = (i == j) ? 1.0f : 0.0f;
    }
    __syncthreads();
    // Gauss-Jordan with partial pivoting on [14 x 28] (torch.linalg.inv: LU with partial pivoting; :201)
    for (int c = 0; c < DX; ++c) {
        if (tid == 0) {
            int best = c;
            float bv = fabsf(M[c * 2 * DX + c]);
            for (int r = c + 1; r < DX; ++r) {
                const float v = fabsf(M[r * 2 * DX + c]);
                if (v > bv) { bv = v; best = r; }
            }
            piv = best;
            if (bv == 0.0f) *p.singular = 1;
        }
        __syncthreads();
        if (piv != c && tid < 2 * DX) {
            const float t = M[c * 2 * DX + tid];
            M[c * 2 * DX + tid] = M[piv * 2 * DX + tid];
            M[piv * 2 * DX + tid] = t;
        }
        __syncthreads();
        const float d = M[c * 2 * DX + c];
        __syncthreads();
        if (tid < 2 * DX) M[c * 2 * DX + tid] = M[c * 2 * DX + tid] / d;
        __syncthreads();
        float f[7];                                             // 14 x 28 = 392 entries over 64 threads: 7 each
#pragma unroll
        for (int u = 0; u < 7; ++u) {
            const int idx = tid + 64 * u, r = idx / (2 * DX);
            f[u] = (idx < DX * 2 * DX && r != c) ? M[r * 2 * DX + c] : 0.0f;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 7; ++u) {
            const int idx = tid + 64 * u, r = idx / (2 * DX), cc = idx - r * 2 * DX;
            if (idx < DX * 2 * DX && r != c) M[idx] -= f[u] * M[c * 2 * DX + cc];
        }
        __syncthreads();
    }
    for (int idx = tid; idx < DX * DX; idx += 64) {     // K = P inv   (:202-204)
        const int i = idx / DX, j = idx - i * DX;
        float a = 0.0f;
        for (int k = 0; k < DX; ++k) a += P[i * DX + k] * M[k * 2 * DX + DX + j];
        Kg[idx] = a;
    }
    __syncthreads();
    // corrected = pred + (K (y - H X))^T   (:206-208); sp holds A = pred - mean
    for (int i = tid; i < E * DX; i += 64) {
        const int e = i / DX, c = i - e * DX;
        float a = 0.0f;
        for (int k = 0; k < DX; ++k) a += Kg[c * DX + k] * sz[e * DX + k];
        const float v = (sp[i] + mp[c]) + a;
        sp[i] = v;
        p.corrected[(size_t)s * E * DX + i] = v;
    }
    __syncthreads();
    if (tid < DX) {                                     // :211
        float a = 0.0f;
        for (int e = 0; e < E; ++e) a += sp[e * DX + tid];
        p.m_corrected[(size_t)s * DX + tid] = a / (float)E;
    }
}

__global__ void kf_format_state_kernel(const float* state, const float* noise, float* out, int S, int E,
                                       unsigned long long seed) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (unsigned)(S * E * DX)) return;
    const int s = i / (E * DX), c = i % DX;
    const float n = noise ? noise[i] : philox_normal(i, 0x300u, seed);
    out[i] = state[s * DX + c] + 0.31622776601683794f * n;        // N(0, 0.1 I): kalman_models.py:168-171
}

struct KfLayer {
    int N = 0, K = 0;
    float *mu_w = nullptr, *mu_b = nullptr, *rho_w = nullptr, *rho_b = nullptr, *d_w = nullptr, *d_b = nullptr;
};

}  // namespace

struct ape_kalman {
    int E = 0, W = 0, device = 0;
    KfLayer layer[NLAYERS];
    void* slab = nullptr;
    float* ws = nullptr;            // per-forward activations, grown on demand
    int ws_S = 0;
    int* singular = nullptr;
    bool has_weights = false;
};

namespace {

void layer_dims(int W, int (&N)[NLAYERS], int (&K)[NLAYERS]) {
    const int n[NLAYERS] = {256, 512, DX, 256, 256, 64, DX, 32, DX};
    const int k[NLAYERS] = {DX * W, 256, 512, RAW * W, 256, 256, 64, DX, 32};
    for (int i = 0; i < NLAYERS; ++i) { N[i] = n[i]; K[i] = k[i]; }
}

size_t blob_floats(int W) {
    int N[NLAYERS], K[NLAYERS];
    layer_dims(W, N, K);
    size_t t = 0;
    for (int i = 0; i < NLAYERS; ++i) t += (size_t)(IS_FLIP[i] ? 2 : 1) * ((size_t)N[i] * K[i] + N[i]);
    return t;
}

int kfail(int code, const char* msg) { return ape_set_error(code, msg); }

hipError_t launch_linear(const ape_kalman* m, int li, const float* X, int x_group, int R, const float* sin_, const float* sout,
                         unsigned long long seed, int act, float* Y, hipStream_t st) {
    const KfLayer& L = m->layer[li];
    KfLinearParams q{};
    q.X = X; q.x_group = x_group; q.R = R; q.K = L.K; q.N = L.N;
    q.Wmu = L.mu_w; q.bmu = L.mu_b; q.Wd = L.d_w; q.bd = L.d_b;
    q.sign_in = sin_; q.sign_out = sout; q.seed = seed; q.layer = li; q.act = act; q.Y = Y;
    const int Kp = (L.K + 15) & ~15;
    const dim3 grid((L.N + 63) / 64, (R + 15) / 16);
    const size_t smem = (size_t)(IS_FLIP[li] ? 2 : 1) * 16 * (Kp + 4) * sizeof(float);
    if (IS_FLIP[li]) hipLaunchKernelGGL(kf_linear_kernel<true>, grid, dim3(256), smem, st, q);
    else hipLaunchKernelGGL(kf_linear_kernel<false>, grid, dim3(256), smem, st, q);
    return hipGetLastError();
}

}  // namespace

extern "C" {

int ape_kalman_create(const ape_kalman_dims_t* dims, ape_kalman_t** out) {
    if (!dims || !out) return kfail(APE_ERR_INVALID_ARG, "kalman_create: NULL argument");
    if (dims->num_ensemble < 2 || dims->num_ensemble > MAXE)
        return kfail(APE_ERR_INVALID_ARG, "kalman_create: num_ensemble must be in [2, 128]");
    if (dims->win_size < 2 || dims->win_size > 22 || dims->win_size % 2 != 0)   // 22 W <= 512 (the LDS tile of kf_linear); 16-byte rows
        return kfail(APE_ERR_INVALID_ARG, "kalman_create: win_size must be even and in [2, 22]");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || dims->device < 0 || dims->device >= n_dev)
        return kfail(APE_ERR_HIP, "kalman_create: no such HIP device (no CPU fallback)");
    if (hipSetDevice(dims->device) != hipSuccess) return kfail(APE_ERR_HIP, "kalman_create: hipSetDevice failed");
    ape_kalman* m = new (std::nothrow) ape_kalman;
    if (!m) return kfail(APE_ERR_HIP, "kalman_create: out of host memory");
    m->E = dims->num_ensemble; m->W = dims->win_size; m->device = dims->device;
    int N[NLAYERS], K[NLAYERS];
    layer_dims(m->W, N, K);
    size_t total = 256;
    for (int i = 0; i < NLAYERS; ++i)
        total += (size_t)(IS_FLIP[i] ? 3 : 1) * (((size_t)N[i] * K[i] + 63) / 64 * 64 + ((size_t)N[i] + 63) / 64 * 64);
    if (hipMalloc(&m->slab, total * sizeof(float)) != hipSuccess || hipMemset(m->slab, 0, total * sizeof(float)) != hipSuccess) {
        delete m;
        return kfail(APE_ERR_HIP, "kalman_create: device allocation failed");
    }
    float* cur = static_cast<float*>(m->slab);
    m->singular = reinterpret_cast<int*>(cur);
    cur += 64;
    for (int i = 0; i < NLAYERS; ++i) {
        KfLayer& L = m->layer[i];
        L.N = N[i]; L.K = K[i];
        const size_t wn = ((size_t)N[i] * K[i] + 63) / 64 * 64, bn = ((size_t)N[i] + 63) / 64 * 64;
        L.mu_w = cur; cur += wn; L.mu_b = cur; cur += bn;
        if (IS_FLIP[i]) {
            L.rho_w = cur; cur += wn; L.rho_b = cur; cur += bn;
            L.d_w = cur; cur += wn; L.d_b = cur; cur += bn;
        }
    }
    *out = m;
    return APE_OK;
}

int ape_kalman_destroy(ape_kalman_t* m) {
    if (!m) return APE_OK;
    (void)hipSetDevice(m->device);
    if (m->slab) (void)hipFree(m->slab);
    if (m->ws) (void)hipFree(m->ws);
    delete m;
    return APE_OK;
}

size_t ape_kalman_weight_floats(const ape_kalman_t* m) { return m ? blob_floats(m->W) : 0; }

size_t ape_kalman_noise_floats(const ape_kalman_t* m, int32_t S) {
    if (!m || S < 1) return 0;
    size_t t = 0;
    const size_t R = (size_t)S * m->E;
    for (int i = 0; i < NLAYERS; ++i)
        if (IS_FLIP[i]) t += (size_t)m->layer[i].N * m->layer[i].K + m->layer[i].N + R * m->layer[i].K + R * m->layer[i].N;
    return t;
}

int ape_kalman_load_weights(ape_kalman_t* m, const float* blob, size_t n_floats) {
    if (!m || !blob) return kfail(APE_ERR_INVALID_ARG, "kalman_load_weights: NULL argument");
    if (n_floats != blob_floats(m->W)) return kfail(APE_ERR_INVALID_ARG, "kalman_load_weights: blob size does not match the model");
    if (hipSetDevice(m->device) != hipSuccess) return kfail(APE_ERR_HIP, "kalman_load_weights: hipSetDevice failed");
    const float* cur = blob;
    for (int i = 0; i < NLAYERS; ++i) {
        KfLayer& L = m->layer[i];
        const size_t wn = (size_t)L.N * L.K;
        hipError_t e = hipMemcpy(L.mu_w, cur, wn * sizeof(float), hipMemcpyHostToDevice); cur += wn;
        if (IS_FLIP[i]) {
            if (e == hipSuccess) e = hipMemcpy(L.rho_w, cur, wn * sizeof(float), hipMemcpyHostToDevice);
            cur += wn;
        }
        if (e == hipSuccess) e = hipMemcpy(L.mu_b, cur, L.N * sizeof(float), hipMemcpyHostToDevice);
        cur += L.N;
        if (IS_FLIP[i]) {
            if (e == hipSuccess) e = hipMemcpy(L.rho_b, cur, L.N * sizeof(float), hipMemcpyHostToDevice);
            cur += L.N;
        }
        if (e != hipSuccess) return kfail(APE_ERR_HIP, "kalman_load_weights: copy to the device failed");
    }
    m->has_weights = true;
    return APE_OK;
}

int ape_kalman_forward(ape_kalman_t* m, const float* raw_obs_dev, const float* state_prev_dev, int32_t S, uint64_t seed,
                       const float* noise_dev, float* state_corrected_dev, float* m_state_corrected_dev,
                       float* m_state_pred_dev, float* z_dev, float* ensemble_z_dev, void* stream) {
    if (!m || !raw_obs_dev || !state_prev_dev || !state_corrected_dev || !m_state_corrected_dev || !m_state_pred_dev || !z_dev ||
        !ensemble_z_dev)
        return kfail(APE_ERR_INVALID_ARG, "kalman_forward: NULL argument");
    if (S < 1 || S > 65535) return kfail(APE_ERR_INVALID_ARG, "kalman_forward: S must be in [1, 65535]");
    if (!m->has_weights) return kfail(APE_ERR_NOT_READY, "kalman_forward: weights not loaded");
    if (hipSetDevice(m->device) != hipSuccess) return kfail(APE_ERR_HIP, "kalman_forward: hipSetDevice failed");
    hipStream_t st = (hipStream_t)stream;
    const int E = m->E, R = S * E;
    if (S > m->ws_S) {                                  // activations: h1 [R,256] h2 [R,512] pred [R,14] s1 [S,256] s2 [R,256] s3 [R,64]
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (st && hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
            return kfail(APE_ERR_CAPACITY, "kalman_forward: the workspace cannot grow during stream capture");
        if (m->ws) { (void)hipFree(m->ws); m->ws = nullptr; m->ws_S = 0; }
        const size_t fl = (size_t)S * E * (256 + 512 + 16 + 256 + 64) + (size_t)S * 256;
        if (hipMalloc((void**)&m->ws, fl * sizeof(float)) != hipSuccess) return kfail(APE_ERR_HIP, "kalman_forward: workspace allocation failed");
        m->ws_S = S;
    }
    float* h1 = m->ws;
    float* h2 = h1 + (size_t)R * 256;
    float* pred = h2 + (size_t)R * 512;
    float* s2 = pred + (size_t)R * 16;
    float* s3 = s2 + (size_t)R * 256;
    float* s1 = s3 + (size_t)R * 64;
    // ---- perturbations of the five flipout layers; injected draws: per layer eps_w, eps_b, sign_in [R,K], sign_out [R,N]
    KfPerturbParams pp{};
    const float* sin_[NLAYERS] = {};
    const float* sout[NLAYERS] = {};
    {
        const float* cur = noise_dev;
        unsigned run = 0;
        int seg = 0;
        for (int i = 0; i < NLAYERS; ++i) {
            if (!IS_FLIP[i]) continue;
            const KfLayer& L = m->layer[i];
            const unsigned wn = (unsigned)(L.N * L.K);
            pp.rho[seg] = L.rho_w; pp.delta[seg] = L.d_w; pp.start[seg] = run; run += wn;
            pp.rho[seg + 1] = L.rho_b; pp.delta[seg + 1] = L.d_b; pp.start[seg + 1] = run; run += (unsigned)L.N;
            if (cur) {
                pp.eps[seg] = cur; cur += wn;
                pp.eps[seg + 1] = cur; cur += L.N;
                sin_[i] = cur; cur += (size_t)R * L.K;
                sout[i] = cur; cur += (size_t)R * L.N;
            }
            seg += 2;
        }
        pp.start[10] = run;
        pp.seed = seed;
        hipLaunchKernelGGL(kf_perturb_kernel, dim3((run + 255) / 256), dim3(256), 0, st, pp);
        if (hipGetLastError() != hipSuccess) return kfail(APE_ERR_HIP, "kalman_forward: perturbation launch failed");
    }
    hipError_t e = launch_linear(m, P_B1, state_prev_dev, 1, R, sin_[P_B1], sout[P_B1], seed, 1, h1, st);
    if (e == hipSuccess) e = launch_linear(m, P_B3, h1, 1, R, sin_[P_B3], sout[P_B3], seed, 1, h2, st);
    if (e == hipSuccess) e = launch_linear(m, P_M2, h2, 1, R, nullptr, nullptr, seed, 0, pred, st);
    if (e == hipSuccess) e = launch_linear(m, S_FC2, raw_obs_dev, 1, S, nullptr, nullptr, seed, 1, s1, st);
    if (e == hipSuccess) e = launch_linear(m, S_FC3, s1, E, R, sin_[S_FC3], sout[S_FC3], seed, 1, s2, st);
    if (e == hipSuccess) e = launch_linear(m, S_FC5, s2, 1, R, sin_[S_FC5], sout[S_FC5], seed, 1, s3, st);
    if (e == hipSuccess) e = launch_linear(m, S_FC6, s3, 1, R, sin_[S_FC6], sout[S_FC6], seed, 0, ensemble_z_dev, st);
    if (e != hipSuccess) return kfail(APE_ERR_HIP, "kalman_forward: layer launch failed");
    KfUpdateParams u{};
    u.pred = pred; u.ensz = ensemble_z_dev;
    u.w1 = m->layer[O_FC1].mu_w; u.b1 = m->layer[O_FC1].mu_b; u.w2 = m->layer[O_FC2].mu_w; u.b2 = m->layer[O_FC2].mu_b;
    u.corrected = state_corrected_dev; u.m_corrected = m_state_corrected_dev; u.m_pred = m_state_pred_dev; u.z = z_dev;
    u.S = S; u.E = E; u.singular = m->singular;
    hipLaunchKernelGGL(kf_update_kernel, dim3(S), dim3(64), 0, st, u);
    if (hipGetLastError() != hipSuccess) return kfail(APE_ERR_HIP, "kalman_forward: update launch failed");
    return APE_OK;
}

int ape_kalman_format_state(ape_kalman_t* m, const float* state_dev, int32_t S, uint64_t seed, const float* noise_dev,
                            float* out_dev, void* stream) {
    if (!m || !state_dev || !out_dev) return kfail(APE_ERR_INVALID_ARG, "kalman_format_state: NULL argument");
    if (S < 1) return kfail(APE_ERR_INVALID_ARG, "kalman_format_state: S must be >= 1");
    if (hipSetDevice(m->device) != hipSuccess) return kfail(APE_ERR_HIP, "kalman_format_state: hipSetDevice failed");
    const unsigned n = (unsigned)(S * m->E * DX);
    hipLaunchKernelGGL(kf_format_state_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, state_dev, noise_dev, out_dev,
                       S, m->E, (unsigned long long)seed);
    if (hipGetLastError() != hipSuccess) return kfail(APE_ERR_HIP, "kalman_format_state: launch failed");
    return APE_OK;
}

int ape_kalman_check(ape_kalman_t* m) {
    if (!m) return kfail(APE_ERR_INVALID_ARG, "kalman_check: NULL model");
    if (hipSetDevice(m->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return kfail(APE_ERR_HIP, "kalman_check: synchronise failed");
    int s = 0;
    if (hipMemcpy(&s, m->singular, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return kfail(APE_ERR_HIP, "kalman_check: copy failed");
    if (s != 0) {
        (void)hipMemset(m->singular, 0, sizeof(int));
        return kfail(APE_ERR_HIP, "kalman forward: a singular innovation matrix was inverted since the last check (torch.linalg.inv raises there)");
    }
    return APE_OK;
}

}  // extern "C"
