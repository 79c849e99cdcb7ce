// C ABI of libape_hip.so (see include/ape_hip.h).  Host side: handle, weight packing, checks,
// kernel dispatch.  No CPU fallback: every compute entry point needs a gfx950 device.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <atomic>
#include <chrono>
#include <cstring>
#include <new>
#include <string>
#include <utility>
#include <vector>

#include "../../include/ape_hip.h"
#include "ape_internal.h"
#include "ape_model.h"

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) return fail(APE_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

int layout_targets(int layout) {
    switch (layout) {
        case APE_LAYOUT_ORI_CAL_LARM_UARM_HIPS: return 14;
        case APE_LAYOUT_ORI_CAL_LARM_UARM: return 12;
        case APE_LAYOUT_ORI_POS_CAL_LARM_UARM_HIPS: return 20;
    }
    return -1;
}
int layout_est_width(int layout) { return layout == APE_LAYOUT_ORI_CAL_LARM_UARM ? 14 : 21; }

int check_dims(const ape_dims_t* d) {
    if (!d) return fail(APE_ERR_INVALID_ARG, "dims is NULL");
    if (d->input_size < 1 || d->input_size > APE_MAX_INPUT)
        return fail(APE_ERR_UNSUPPORTED, "input_size %d outside 1..%d", d->input_size, APE_MAX_INPUT);
    if (d->hidden_size != 128 && d->hidden_size != 256)
        return fail(APE_ERR_UNSUPPORTED, "hidden_size %d: kernels are built for 128 and 256", d->hidden_size);
    if (d->model_kind != APE_MODEL_LSTM && d->model_kind != APE_MODEL_FF && d->model_kind != APE_MODEL_IMUPOSE)
        return fail(APE_ERR_INVALID_ARG, "unknown model_kind %d", d->model_kind);
    if (d->model_kind == APE_MODEL_IMUPOSE && (d->hidden_size != 256 || d->num_layers != 2))
        return fail(APE_ERR_INVALID_ARG, "ImuPoseLSTM is a fixed 2 x 256 LSTM behind a 256-wide input layer "
                    "(nn_models.py:222-229), got hidden_size %d num_layers %d", d->hidden_size, d->num_layers);
    if (d->model_kind != APE_MODEL_FF && (d->num_layers < 1 || d->num_layers > APE_MAX_LAYERS))
        return fail(APE_ERR_UNSUPPORTED, "num_layers %d outside 1..%d", d->num_layers, APE_MAX_LAYERS);
    if (d->model_kind == APE_MODEL_FF && (d->num_layers < 0 || d->num_layers > APE_MAX_FF_LAYERS - 1))
        return fail(APE_ERR_UNSUPPORTED, "hidden_layer_count %d outside 0..%d", d->num_layers, APE_MAX_FF_LAYERS - 1);
    if (d->output_size < 1 || d->output_size > APE_MAX_OUTPUT)
        return fail(APE_ERR_UNSUPPORTED, "output_size %d outside 1..%d", d->output_size, APE_MAX_OUTPUT);
    if (d->target_layout == APE_LAYOUT_NONE) return APE_OK;      // regressor only, no post-filter
    const int want = layout_targets(d->target_layout);
    if (want < 0) return fail(APE_ERR_INVALID_ARG, "unknown target_layout %d", d->target_layout);
    if (want != d->output_size)
        return fail(APE_ERR_INVALID_ARG, "target_layout %d needs output_size %d, got %d", d->target_layout, want,
                    d->output_size);
    return APE_OK;
}

int padded_input(int I) { return ((I + 31) / 32) * 32; }

bool parse_kind_dims(int kind, int* width, int* I) {
    switch (kind) {
        case APE_PARSE_WATCH_PHONE_POCKET: *width = 55; *I = 22; return true;
        case APE_PARSE_WATCH_ONLY: *width = 28; *I = 20; return true;
        case APE_PARSE_WATCH_ONLY_PHONE_MSG: *width = 55; *I = 20; return true;
        case APE_PARSE_WATCH_PHONE_UARM: *width = 55; *I = 38; return true;
    }
    return false;
}

}  // namespace

// ---- how a batch is split over the device's CUs (pure arithmetic: unit-tested on the CPU via ape_debug_plan) ----
// A cluster is GH = H/16 workgroups, one per CU, so a device with n_cus CUs runs n_cus / GH clusters at once
// (16 on a whole MI355X at H = 256); a batch-tile "wave" is one 16-row workgroup per CU (4096 rows on 256 CUs).
static int cluster_capacity(int n_cus, int H) { return n_cus / (H / 16); }
static int tile16_wave_rows(int n_cus) { return APE_TILE_ROWS * n_cus; }
// smallest row-tile count (16 rows each) per cluster that fits `rows` into one launch; the dropout variants are
// built for at most 2 tiles
static int cluster_nmt(int n_cus, int H, int rows, bool cdrop) {
    const int cap = cluster_capacity(n_cus, H);
    for (int cand : {1, 2, 4})
        if ((!cdrop || cand <= 2) && (rows + 16 * cand - 1) / (16 * cand) <= cap) return cand;
    return cdrop ? 2 : 4;
}
// fp16 v2 kernel: 8-member clusters of 32 rows, formed within the 8 block-index classes, so a launch carries whole
// groups of 8 clusters = 64 workgroups, all of which must be able to be resident together
static int f16v2_capacity(int n_cus) { return (n_cus / 64) * 8; }
static int cluster_rows_per_launch(int n_cus, int H, bool cdrop) { return 16 * (cdrop ? 2 : 4) * cluster_capacity(n_cus, H); }

// APE_KERNEL_AUTO: how many whole waves of the batch-tile kernel to peel off the front of a batch.  Measured on a
// whole MI355X (microseconds): a batch-tile wave sustains 125 TFLOP/s at H = 256 and 109 at H = 128 whatever T and
// the dropout mode (a partial wave costs a whole one); a cluster launch costs 25 + 13.7 T (12.5 + 8.3 T for the 2-tile
// dropout variant with XCD-local clusters) however few of its rows are used.  Both rates scale with the CU count of the device.
// `wide` (ImuPoseLSTM, 256-wide layer-0 input): a full batch-tile wave sustains 123 TFLOP/s, the two-tile cluster launch
// (512 rows) costs 20 + 11 T -- 95 TFLOP/s when full, so whole waves go to the batch-tile kernel and the rest to the cluster.
// which cluster kernel serves `rest` rows behind the batch-tile waves (the ONE rule lstm_forward_impl, the cost model and
// ape_debug_plan share): the second-generation f32 kernel from 513 rows on where the model and the call allow it (`c32`:
// a 2 x 256 model, eval mode, last-step output), else the first-generation kernel
// `gen2`: the second-generation kernel the model and the call are eligible for -- 32: lstm_cluster32.hip (2 x 256), 16:
// lstm_cluster16.hip (3 x 128: a launch costs 24.6 + 6.6 T against the first generation's 15.5 + 7.5 T at 513 .. 1024 rows -- with
// its XCD-local clusters; 15.1 + 7.75 T before them --, so it serves windows of 12 steps and more), 0: none
// (round 6) 48 = 16 + the level-synchronous kernel lstm_level16.hip: T + 2 hand-overs per launch instead of a three-layer pipeline with four
// fill / drain phases.  It serves ONE launch's worth of rows: 5 .. 512 with one row tile per cluster at every window length, 513 .. 1024 with
// two row tiles per cluster (two agents per workgroup) up to 48 steps; lstm_cluster16.hip keeps the longer windows and the larger batches
enum { PLAN_NONE = 0, PLAN_GEN1 = 1, PLAN_C32 = 2, PLAN_SMALL = 3, PLAN_C16 = 4, PLAN_LV16 = 5 };
#define APE_LV16_MIN_ROWS 5          // (up to 4 rows: the latency kernel)
#define APE_LV16_MAX_T 48           // two row tiles per cluster, 513 .. 1024 rows: 170.6 / 222.2 / 327.8 us at 24 / 32 / 48 steps against
                                    // lstm_cluster16.hip's 179.5 / 231.7 / 339.8; a tie at 64
#define APE_LV16_MAX_T_SINGLE 4000  // one row tile per cluster, up to 512 rows: faster than the first generation at every window measured
                                    // (512 rows: 40.0 / 66.9 / 123.2 / 231.5 / 304.9 us at 6 / 12 / 24 / 48 / 64 steps against 42.8 / 70.4 / 129.7 /
                                    // 245.3 / 323.2; 5 rows: 34.3 against 42.8); a level's tag holds 12 bits of step count
#define APE_LV16_COST_US0 14.0      // a launch of up to 1024 rows: microseconds = US0 + US_T x T (measured)
#define APE_LV16_COST_US_T 6.8
#define APE_LV16_COST1_US0 13.0     // ... of up to 512 rows (one row tile per cluster)
#define APE_LV16_COST1_US_T 4.5
static int rest_kernel(int rest, int T, int gen2, int n_cus) {
    if (rest <= 0) return PLAN_NONE;
    if (gen2 == 32 && rest > 512) return PLAN_C32;
    static const int c16_min_t = getenv("APE_C16_MIN_T") ? atoi(getenv("APE_C16_MIN_T")) : 12;      // (diagnostic overrides for A/B runs)
    static const int lv16_max_t = getenv("APE_LV16_MAX_T") ? atoi(getenv("APE_LV16_MAX_T")) : APE_LV16_MAX_T;
    static const int lv16_min_rows = getenv("APE_LV16_MIN_ROWS") ? atoi(getenv("APE_LV16_MIN_ROWS")) : APE_LV16_MIN_ROWS;
    // (ONE launch only: its workgroups take a CU's whole LDS, so a second launch cannot start under the first one's tail as the first
    //  generation's do -- 2048 x 6: 108 us in two launches against 100)
    if (gen2 == 48 && rest >= lv16_min_rows) {
        if (rest <= 16 * ape_level16_max_clusters(n_cus) && T <= APE_LV16_MAX_T_SINGLE) return PLAN_LV16;       // one row tile per cluster
        if (rest <= 32 * ape_level16_max_clusters(n_cus) && T <= lv16_max_t) return PLAN_LV16;
    }
    if ((gen2 == 16 || gen2 == 48) && rest > 512 && T >= c16_min_t) return PLAN_C16;
    return PLAN_GEN1;
}
static int gen2_of(const ape_model* m) {
    return (m->c32_ok && m->c32_on) ? 32 : (m->c16_ok && m->c32_on) ? (m->lv16_ok ? 48 : 16) : 0;
}

static int auto_tile16_waves(const ape_dims_t* dims, int n_cus, int B, int T, bool cdrop, int gen2 = 0, bool wide = false) {
    const int wave = tile16_wave_rows(n_cus), rpl = cluster_rows_per_launch(n_cus, dims->hidden_size, cdrop || wide);
    if (rpl == 0) return (B + wave - 1) / wave;          // no cluster fits on this device
    const double rate = (wide ? 1.23e14 : dims->hidden_size == 256 ? 1.25e14 : 1.09e14) * n_cus / 256.0;
    const double t16 = (double)wave * ape_flops_per_window(dims, T) / rate * 1e6;
    // first-generation launches: 25 + 13.7 T (12.5 + 8.3 T with dropout, 20 + 11 T wide); second-generation f32 kernel, eval mode:
    // 16 + 12.4 T per launch of up to 32 x f16v2_capacity rows -- priced only where rest_kernel() really picks it
    // (round 4: the dropout form is priced at what it measures with XCD-local clusters, 12.5 + 8.3 T -- the Philox counters name global
    //  rows in every kernel now, so the route a Monte-Carlo call takes no longer decides which samples it draws)
    const double tcl1 = wide ? 20.0 + 11.0 * T : cdrop ? 12.5 + 8.3 * T : 25.0 + 13.7 * T;
    const int rpl2 = 32 * f16v2_capacity(n_cus);
    auto cost = [&](int w) {
        const int rest = B - wave * w;
        if (rest <= 0) return w * t16;
        const int k = rest_kernel(rest, T, (!cdrop && !wide && rpl2 > 0) ? gen2 : 0, n_cus);
        if (k == PLAN_C32) return w * t16 + (double)((rest + rpl2 - 1) / rpl2) * (16.0 + 12.4 * T);
        if (k == PLAN_C16) return w * t16 + (double)((rest + rpl2 - 1) / rpl2) * (24.6 + 6.6 * T);
        if (k == PLAN_LV16) {
            if (rest <= 16 * ape_level16_max_clusters(n_cus)) return w * t16 + APE_LV16_COST1_US0 + APE_LV16_COST1_US_T * T;
            return w * t16 + APE_LV16_COST_US0 + APE_LV16_COST_US_T * T;
        }
        return w * t16 + (double)((rest + rpl - 1) / rpl) * tcl1;
    };
    int best = 0;
    double best_cost = cost(0);
    for (int w : {B / wave, (B + wave - 1) / wave})
        if (w > 0 && cost(w) < best_cost) { best = w; best_cost = cost(w); }
    return best;
}


int ape_set_error(int code, const char* msg) { return fail(code, "%s", msg); }

// every successfully enqueued compute call of a handle is remembered until its next successful check (ape_model_recover)
static void journal_add(ape_model* m, const ApeJournalEntry& e) {
    if (!m || m->replaying) return;
    if (m->journal_n < APE_JOURNAL_CAP) m->journal[m->journal_n++] = e;
    else m->journal_overflow = true;
}
static void journal_clear(ape_model* m) { m->journal_n = 0; m->journal_overflow = false; }

extern "C" {

int ape_abi_version(void) { return APE_ABI_VERSION; }
const char* ape_last_error(void) { return g_err.c_str(); }

int ape_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    int ok = 0;
    for (int i = 0; i < n; ++i) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, i) == hipSuccess && strncmp(prop.gcnArchName, "gfx950", 6) == 0) ++ok;
    }
    return ok;
}

size_t ape_weight_blob_floats(const ape_dims_t* d) {
    if (check_dims(d) != APE_OK) return 0;
    const size_t H = d->hidden_size, I = d->input_size, O = d->output_size;
    if (d->model_kind == APE_MODEL_FF) return H * I + H + (size_t)d->num_layers * (H * H + H) + O * H + O;
    size_t n = 0;
    const size_t in0 = (d->model_kind == APE_MODEL_IMUPOSE) ? H : I;          // LSTM layer 0 reads the input layer's output
    if (d->model_kind == APE_MODEL_IMUPOSE) n += H * I + H;
    for (int l = 0; l < d->num_layers; ++l) n += 4 * H * (l == 0 ? in0 : H) + 4 * H * H + 8 * H;
    return n + O * H + O;
}

double ape_flops_per_window(const ape_dims_t* d, int32_t T) {
    if (check_dims(d) != APE_OK || T < 1) return 0.0;
    const double H = d->hidden_size, I = d->input_size, O = d->output_size;
    if (d->model_kind == APE_MODEL_FF) return 2.0 * (I * H + d->num_layers * H * H + O * H);   // last step only
    double step = 0;
    const double in0 = (d->model_kind == APE_MODEL_IMUPOSE) ? H : I;
    if (d->model_kind == APE_MODEL_IMUPOSE) step += 2.0 * I * H;
    for (int l = 0; l < d->num_layers; ++l) step += 2.0 * 4.0 * H * ((l == 0 ? in0 : H) + H);
    return step * T + 2.0 * O * H;
}

int ape_model_create(const ape_dims_t* dims, ape_model_t** out) {
    if (!out) return fail(APE_ERR_INVALID_ARG, "out_model is NULL");
    *out = nullptr;
    if (int rc = check_dims(dims)) return rc;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) {
        (void)hipGetLastError();
        return fail(APE_ERR_NO_DEVICE, "no HIP device visible: libape_hip has no CPU fallback");
    }
    if (dims->device < 0 || dims->device >= n) return fail(APE_ERR_INVALID_ARG, "device %d of %d", dims->device, n);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, dims->device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(APE_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 only", dims->device,
                    prop.gcnArchName);
    HIP_TRY(hipSetDevice(dims->device));

    ape_model* m = new (std::nothrow) ape_model();
    if (!m) return fail(APE_ERR_HIP, "out of host memory");
    m->dims = *dims;
    m->n_cus = prop.multiProcessorCount;            // 256 on a whole MI355X; fewer on a partitioned / CU-masked device
    const bool imupose = dims->model_kind == APE_MODEL_IMUPOSE;
    m->lstm_in = imupose ? dims->hidden_size : dims->input_size;
    m->KX = padded_input(m->lstm_in);
    m->KXpre = padded_input(dims->input_size);
    const double def_body[9] = {-0.22, 0, 0, -0.26, 0, 0, -0.1704612, 0.4309841, -0.00670862};  // bone_map.py:42-45
    memcpy(m->body, def_body, sizeof(def_body));
    const int H = dims->hidden_size, L = dims->num_layers, O = dims->output_size, I = dims->input_size;
    const int NT = 4 * (H / 64);
    // per-device kernel attributes of the MLP / head-rows kernels (every model kind can reach one of them)
    hipError_t e = ape_prepare_mlp_tile16(H);
    // every fixed-size device buffer of the model comes out of ONE allocation (planned first, carved after at
    // 256-byte boundaries, zero-filled): one driver call to create, one to free, and the flag / ticket words start
    // at zero without separate memsets
    std::vector<std::pair<void**, size_t>> plan_list;
    auto plan = [&](void** ptr, size_t bytes) { plan_list.emplace_back(ptr, bytes); return hipSuccess; };
    auto commit_plan = [&]() -> hipError_t {
        size_t total = 0;
        for (auto& it : plan_list) total += (it.second + 255) / 256 * 256;
        hipError_t ee = hipMalloc(&m->slab, total);
        if (ee != hipSuccess) return ee;
        m->slab_bytes = total;
        ee = hipMemset(m->slab, 0, total);
        if (ee != hipSuccess) return ee;
        size_t off = 0;
        for (auto& it : plan_list) { *it.first = static_cast<char*>(m->slab) + off; off += (it.second + 255) / 256 * 256; }
        return hipSuccess;
    };
    if (dims->model_kind == APE_MODEL_FF) {
        for (int j = 0; j <= L && e == hipSuccess; ++j) {
            const size_t K = (j == 0) ? m->KX : H;
            e = plan((void**)&m->ff_wpack[j], K * H * sizeof(float));
            if (e == hipSuccess) e = plan((void**)&m->ff_bias[j], H * sizeof(float));
        }
        if (e == hipSuccess) e = plan((void**)&m->w_out, (size_t)O * H * sizeof(float));
        if (e == hipSuccess) e = plan((void**)&m->b_out, O * sizeof(float));
        if (e == hipSuccess) e = plan((void**)&m->stats, (3 * I + 2 * O) * sizeof(double));
        if (e == hipSuccess && ape_mlp_pipe_supported(H, L, m->KX, O) && m->n_cus >= 16) {
            e = plan((void**)&m->ffp_wa0, (size_t)4 * 2 * 4 * 64 * 4 * sizeof(float));
            if (e == hipSuccess) e = plan((void**)&m->ffp_wa1, (size_t)4 * 2 * 32 * 64 * 4 * sizeof(float));
            if (e == hipSuccess) e = plan((void**)&m->ffp_wb2, (size_t)4 * 2 * 32 * 64 * 4 * sizeof(float));
            if (e == hipSuccess) e = plan((void**)&m->ffp_wbo, (size_t)4 * 8 * 64 * 4 * sizeof(float));
            m->ffp_ring_bytes = ape_mlp_pipe_ring_bytes(m->n_cus);
            m->ffp_ctl_words = ape_mlp_pipe_ctl_words(m->n_cus);
            if (e == hipSuccess) e = plan((void**)&m->ffp_ring, m->ffp_ring_bytes);
            if (e == hipSuccess) e = plan((void**)&m->ffp_ctl, m->ffp_ctl_words * sizeof(unsigned));
            if (e == hipSuccess) e = ape_prepare_mlp_pipe();
            m->ffp_ok = e == hipSuccess;
        }
        if (e == hipSuccess) e = commit_plan();
        if (e != hipSuccess) {
            ape_model_destroy(m);
            return fail(APE_ERR_HIP, "model allocation failed: %s", hipGetErrorString(e));
        }
        m->kernel_name = H == 256 ? "ape_mlp_tile16<256>" : "ape_mlp_tile16<128>";
        *out = m;
        return APE_OK;
    }
    for (int l = 0; l < L && e == hipSuccess; ++l) {
        const size_t Q = ((l == 0 ? m->KX : H) + H) / 16;
        e = plan((void**)&m->wpack[l], 4 * Q * NT * 64 * sizeof(f32x4));
        if (e == hipSuccess) e = plan((void**)&m->bias[l], 4 * H * sizeof(float));
    }
    if (e == hipSuccess) e = plan((void**)&m->w_out, (size_t)O * H * sizeof(float));
    if (e == hipSuccess) e = plan((void**)&m->b_out, O * sizeof(float));
    if (e == hipSuccess) e = plan((void**)&m->stats, (3 * I + 2 * O) * sizeof(double));
    if (imupose) {       // input layer: the MLP kernel's packing and launch, with the activation as its result
        if (e == hipSuccess) e = plan((void**)&m->ff_wpack[0], (size_t)m->KXpre * H * sizeof(float));
        if (e == hipSuccess) e = plan((void**)&m->ff_bias[0], H * sizeof(float));
        if (e == hipSuccess) e = ape_prepare_lstm_tile16_wide(ape_lstm_tile16_smem_bytes(H, L, m->KX, O, false));
    }
    // the kernel may carve its largest LDS layout (with dropout buffers)
    const size_t smem = ape_lstm_tile16_smem_bytes(H, L, m->KX, O, true);
    if (imupose) {}
    else if (e == hipSuccess && smem <= 160 * 1024) e = ape_prepare_lstm_tile16(H, L, smem);
    else if (e == hipSuccess) e = ape_prepare_lstm_tile16(H, L, ape_lstm_tile16_smem_bytes(H, L, m->KX, O, false));
    if (e != hipSuccess) {
        ape_model_destroy(m);
        return fail(APE_ERR_HIP, "model allocation failed: %s", hipGetErrorString(e));
    }
    if (!imupose && ((H == 256 && L == 2) || (H == 128 && L == 3))) {     // the deployed shapes: layers 1.. can run on their own
        e = ape_prepare_lstm_tile16_upper(H, L - 1, ape_lstm_tile16_smem_bytes(H, L - 1, H, O, true));
        if (e != hipSuccess) {
            ape_model_destroy(m);
            return fail(APE_ERR_HIP, "model allocation failed: %s", hipGetErrorString(e));
        }
        m->upper_ok = true;
    }
    char nm[64];
    snprintf(nm, sizeof(nm), "ape_lstm_tile16<%d, %d, %d>", H, L, imupose ? 16 : 4);
    m->kernel_name = nm;
    snprintf(nm, sizeof(nm), "ape_lstm_cluster<%d, %d, %d", H, L, m->KX);
    m->cluster_name = nm;
    // the cluster kernels need one whole cluster (GH workgroups, one per CU) resident at once: a device with fewer
    // CUs than that gets the batch-tile kernel only (set_kernel(CLUSTER) then answers APE_ERR_UNSUPPORTED)
    if (ape_cluster_supported(H, L, m->KX) && cluster_capacity(m->n_cus, H) >= 1) {
        const int GH = H / 16, max_clusters = cluster_capacity(m->n_cus, H);
        for (int l = 0; l < L && e == hipSuccess; ++l)
            e = plan((void**)&m->wcl[l], (size_t)4 * H * ((l == 0 ? m->KX : H) + H) * sizeof(float));
        m->hx_bytes = (size_t)max_clusters * L * 2 * GH * 64 * 16 * sizeof(float);
        // (upper-layer kernel: 32-row clusters x 2 sets x 2 parities x 32 KB)
        if (ape_upper32_supported(H, L, O) && (size_t)f16v2_capacity(m->n_cus) * 2 * 2 * 32768 > m->hx_bytes)
            m->hx_bytes = (size_t)f16v2_capacity(m->n_cus) * 2 * 2 * 32768;
        // one flag per (cluster, layer, member, wave) -- or, fp16 v2 kernel, per (32-row cluster, row set, member wave) --
        // + the ticket / departure words
        size_t flag_words = (size_t)max_clusters * L * GH * 4;
        if (ape_cluster_f16v2_supported(H, L, m->KX) && (size_t)f16v2_capacity(m->n_cus) * 2 * 64 > flag_words)
            flag_words = (size_t)f16v2_capacity(m->n_cus) * 2 * 64;      // (64 flags per cluster and set in the 16-member form)
        // (second-generation kernel of the 3 x 128 model: one flag per (32-row cluster, layer, member, one of eight waves))
        if (ape_cluster16_supported(H, L, m->KX) && (size_t)f16v2_capacity(m->n_cus) * L * 64 > flag_words)
            flag_words = (size_t)f16v2_capacity(m->n_cus) * L * 64;
        if (ape_upper32_supported(H, L, O) && (size_t)f16v2_capacity(m->n_cus) * 2 * 32 > flag_words)
            flag_words = (size_t)f16v2_capacity(m->n_cus) * 2 * 32;      // (lstm_upper32.hip: one flag per (cluster, set, member wave))
        if (ape_upper128_supported(H, L, O) && !imupose) {
            const int c128 = (m->n_cus / 4) / 8 * 8;
            if (ape_upper128_flag_words(c128) > flag_words) flag_words = ape_upper128_flag_words(c128);
            if (ape_upper128_hx_bytes(c128) > m->hx_bytes) m->hx_bytes = ape_upper128_hx_bytes(c128);
        }
        m->xflag_bytes = ((flag_words * sizeof(unsigned)) + 15) / 16 * 16 + 16;
        if (e == hipSuccess) e = plan((void**)&m->hx, m->hx_bytes);
        if (e == hipSuccess) e = plan((void**)&m->dbg_wg, 256 * 8 * sizeof(unsigned long long));
        if (e == hipSuccess) e = plan((void**)&m->xflags, m->xflag_bytes + 256);
        if (e == hipSuccess) e = plan((void**)&m->xcc_slots, APE_XCC_WORDS * sizeof(unsigned));
        m->hxs_bytes = (size_t)L * 2 * 4 * H * 8;
        if (e == hipSuccess) e = plan((void**)&m->hxs, 256 + m->hxs_bytes);
        m->wide_cluster = imupose;
        for (int l = 0; l < L && e == hipSuccess && !imupose; ++l)
            e = plan(&m->wcl16[l], (size_t)4 * H * ((l == 0 ? m->KX : H) + H) * sizeof(_Float16));
        if (e == hipSuccess) e = ape_prepare_lstm_cluster(H, L, m->KX);
        if (e == hipSuccess && ((H == 128 && L == 3) || (H == 256 && L == 2 && m->KX == 32)))
            e = ape_prepare_lstm_cluster(H, 1, m->KX);       // layer 0 alone: launch A of a Monte-Carlo bank
        if (e == hipSuccess && !imupose) e = ape_prepare_lstm_cluster_f16(H, L, m->KX);
        if (e == hipSuccess && !imupose) e = ape_prepare_lstm_cluster_f16v2(H, L, m->KX);
        // latency kernel with H/8 members (every CU of a 32-CU XCD at H = 256): only where an XCD has that many CUs
        if (m->n_cus / 8 >= H / 8 && !imupose) {
            for (int l = 0; l < L && e == hipSuccess; ++l)
                e = plan((void**)&m->wcls[l], (size_t)4 * H * ((l == 0 ? m->KX : H) + H) * sizeof(float));
            m->small_uw = 2;
        }
        // Monte-Carlo latency kernel (lstm_mc_small.hip): 8 clusters of H/8 members, one per XCD, layer 0 on the latency kernel's
        // H/8-member register image
        if (ape_mc_small_supported(H, L, m->KX) && m->small_uw == 2 && m->n_cus >= 8 * (H / 8) && !imupose) {
            for (int l = 1; l < L && e == hipSuccess; ++l) e = plan((void**)&m->wmc[l], (size_t)8 * H * H * sizeof(float));
            for (int l = 1; l < L && e == hipSuccess; ++l) e = plan((void**)&m->wmc16[l], (size_t)8 * H * H * sizeof(float));
            m->gxm_cluster_bytes = ape_mc_small_cluster_bytes(H, L);
            if (e == hipSuccess) e = plan((void**)&m->gxm, 256 + 8 * m->gxm_cluster_bytes + 8 * 64 * 8);      // + the feature granules of 8 streams
            if (e == hipSuccess) e = ape_prepare_lstm_mc_small(H, L, m->KX);
            m->mcs_ok = true;
        }
        if (ape_cluster32_supported(H, L, m->KX) && f16v2_capacity(m->n_cus) > 0) {
            for (int l = 0; l < L && e == hipSuccess; ++l)
                e = plan((void**)&m->wcl32[l], (size_t)4 * H * ((l == 0 ? m->KX : H) + H) * sizeof(float));
            if (e == hipSuccess) e = ape_prepare_lstm_cluster32(H, L, m->KX);
            m->c32_ok = true;
            if (ape_upper32_supported(H, L, O) && !imupose) {       // shares the layer-1 register image, the exchange buffer and the flags
                if (e == hipSuccess) e = ape_prepare_lstm_upper32();
                m->up32_ok = true;
            }
        }
        // ImuPoseLSTM (2 x 256 behind a 256-wide input layer): each layer is exactly one register image of lstm_upper32.hip's clusters
        // (K = 256 + 256); above 512 windows the LSTM runs there, one layer per launch (lstm_forward_impl)
        if (imupose && H == 256 && L == 2 && m->KX == 256 && ape_upper32_supported(H, L, O) && f16v2_capacity(m->n_cus) >= 8) {
            for (int l = 0; l < L && e == hipSuccess; ++l)
                e = plan((void**)&m->wcl32[l], (size_t)4 * H * (H + H) * sizeof(float));
            if (e == hipSuccess) e = ape_prepare_lstm_upper32();
            m->split32_ok = (e == hipSuccess);
        }
        // the Monte-Carlo bank's weight-stationary route for the 3 x 128 model (lstm_upper128.hip): four-member clusters, whole classes of 8
        if (ape_upper128_supported(H, L, O) && (m->n_cus / 4) / 8 * 8 >= 8 && !imupose) {
            for (int l = 1; l < L && e == hipSuccess; ++l) e = plan((void**)&m->wup128[l], (size_t)4 * H * 2 * H * sizeof(float));
            if (e == hipSuccess) e = ape_prepare_lstm_upper128();
            m->up128_ok = true;
        }
        if (ape_cluster16_supported(H, L, m->KX) && f16v2_capacity(m->n_cus) > 0 && !imupose) {
            if (e == hipSuccess) e = ape_prepare_lstm_cluster16(H, L, m->KX);
            m->c16_ok = true;
            // short windows: the level-synchronous kernel, if the device holds two of its workgroups per CU (else the first generation serves them)
            if (e == hipSuccess && ape_level16_supported(H, L, m->KX) && ape_level16_max_clusters(m->n_cus) > 0) {
                m->gx16_bytes = ape_level16_gx_bytes(m->n_cus);
                e = plan((void**)&m->gx16, 256 + m->gx16_bytes);
                if (e == hipSuccess) e = ape_prepare_lstm_level16(H, L, m->KX);
                m->lv16_ok = (e == hipSuccess);
            }
        }
        if (e != hipSuccess) {
            ape_model_destroy(m);
            return fail(APE_ERR_HIP, "cluster kernel set-up failed: %s", hipGetErrorString(e));
        }
        m->cluster_ok = true;
    }
    e = commit_plan();
    if (e != hipSuccess) {
        ape_model_destroy(m);
        return fail(APE_ERR_HIP, "model allocation failed: %s", hipGetErrorString(e));
    }
    *out = m;
    return APE_OK;
}

int ape_model_destroy(ape_model_t* m) {
    if (!m) return APE_OK;
    (void)hipSetDevice(m->dims.device);
    if (m->slab) (void)hipFree(m->slab);             // weights, packs, statistics, exchange buffers, flags: one allocation
    if (m->y_ws) (void)hipFree(m->y_ws);             // workspaces grow on demand and are their own allocations
    if (m->z_ws) (void)hipFree(m->z_ws);
    if (m->zfrag_ws) (void)hipFree(m->zfrag_ws);
    if (m->hfrag_ws) (void)hipFree(m->hfrag_ws);
    if (m->ypart_ws) (void)hipFree(m->ypart_ws);
    if (m->hseq_ws) (void)hipFree(m->hseq_ws);
    delete m;
    return APE_OK;
}

int ape_model_reserve(ape_model_t* m, int32_t max_batch) {
    if (!m || max_batch < 1) return fail(APE_ERR_INVALID_ARG, "reserve: bad arguments");
    if (max_batch <= m->y_cap) return APE_OK;
    HIP_TRY(hipSetDevice(m->dims.device));
    if (m->y_ws) { HIP_TRY(hipFree(m->y_ws)); m->y_ws = nullptr; m->y_cap = 0; }
    HIP_TRY(hipMalloc((void**)&m->y_ws, (size_t)max_batch * m->dims.output_size * sizeof(float)));
    m->y_cap = max_batch;
    return APE_OK;
}

// Pack [W_ih | W_hh] of one layer into the order the kernel's waves stream it:
//   packed[((w*Q + q)*NT + n)*64 + lane][j] = Wcat[gate*H + w*H/4 + u*16 + (lane&15)][16q + 4(lane>>4) + j]
// with n = gate*UB + u, Wcat columns = input part zero-padded to KXl, then the recurrent part.
int ape_model_load_weights(ape_model_t* m, const float* blob, size_t n_floats) {
    if (!m || !blob) return fail(APE_ERR_INVALID_ARG, "load_weights: NULL argument");
    const size_t want = ape_weight_blob_floats(&m->dims);
    if (n_floats != want) return fail(APE_ERR_INVALID_ARG, "weight blob has %zu floats, expected %zu", n_floats, want);
    HIP_TRY(hipSetDevice(m->dims.device));
    std::vector<float> host(n_floats);
    HIP_TRY(hipMemcpy(host.data(), blob, n_floats * sizeof(float), hipMemcpyDefault));   // host or device source

    const int H = m->dims.hidden_size, L = m->dims.num_layers, O = m->dims.output_size, I = m->dims.input_size;
    const int UB = H / 64, NT = 4 * UB;
    const float* cur = host.data();
    if (m->dims.model_kind == APE_MODEL_FF || m->dims.model_kind == APE_MODEL_IMUPOSE) {
        // per layer [wave 4][k-block q][tile n][lane][4]: lane holds W[w*H/4 + n*16 + (lane&15)][16q + 4(lane>>4) + j]
        const int NTF = H / 64;
        const bool pre_only = m->dims.model_kind == APE_MODEL_IMUPOSE;      // just the input layer, then the LSTM below
        for (int j = 0; j <= (pre_only ? 0 : L); ++j) {
            const int in_j = (j == 0) ? I : H, K = (j == 0) ? (pre_only ? m->KXpre : m->KX) : H, Q = K / 16;
            const float* wj = cur;  cur += (size_t)H * in_j;
            const float* bj = cur;  cur += H;
            std::vector<float> packed((size_t)K * H);
            for (int w = 0; w < 4; ++w)
                for (int q = 0; q < Q; ++q)
                    for (int n = 0; n < NTF; ++n)
                        for (int lane = 0; lane < 64; ++lane) {
                            const int col = w * (H / 4) + n * 16 + (lane & 15);
                            for (int jj = 0; jj < 4; ++jj) {
                                const int k = 16 * q + 4 * (lane >> 4) + jj;
                                packed[((((size_t)w * Q + q) * NTF + n) * 64 + lane) * 4 + jj] =
                                    (k < in_j) ? wj[(size_t)col * in_j + k] : 0.0f;
                            }
                        }
            HIP_TRY(hipMemcpy(m->ff_wpack[j], packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(m->ff_bias[j], bj, H * sizeof(float), hipMemcpyHostToDevice));
        }
        if (!pre_only && m->ffp_ok) {
            // mlp_pipe.hip: v_mfma_f32_32x32x2_f32 A fragments, register 4 kb + j of lane (unit column = lane & 31, hh = lane >> 5) =
            // W[64 wave + 32 ct + (lane & 31)][8 kb + 4 hh + j]; [wave][ct][kb][lane][4]
            const float* wl[3];
            int inl[3], kbl[3];
            const float* c2 = host.data();
            for (int j = 0; j <= 2; ++j) { inl[j] = (j == 0) ? I : H; kbl[j] = (j == 0) ? m->KX / 8 : H / 8; wl[j] = c2; c2 += (size_t)H * inl[j] + H; }
            const float* w_out_h = c2;
            float* dst[3] = {m->ffp_wa0, m->ffp_wa1, m->ffp_wb2};
            for (int j = 0; j <= 2; ++j) {
                std::vector<float> pk((size_t)4 * 2 * kbl[j] * 64 * 4);
                for (int w = 0; w < 4; ++w)
                    for (int ct = 0; ct < 2; ++ct)
                        for (int kb = 0; kb < kbl[j]; ++kb)
                            for (int lane = 0; lane < 64; ++lane)
                                for (int jj = 0; jj < 4; ++jj) {
                                    const int unit = 64 * w + 32 * ct + (lane & 31), k = 8 * kb + 4 * (lane >> 5) + jj;
                                    pk[((((size_t)w * 2 + ct) * kbl[j] + kb) * 64 + lane) * 4 + jj] = (k < inl[j]) ? wl[j][(size_t)unit * inl[j] + k] : 0.0f;
                                }
                HIP_TRY(hipMemcpy(dst[j], pk.data(), pk.size() * sizeof(float), hipMemcpyHostToDevice));
            }
            std::vector<float> po((size_t)4 * 8 * 64 * 4);
            for (int w = 0; w < 4; ++w)
                for (int kb = 0; kb < 8; ++kb)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int jj = 0; jj < 4; ++jj) {
                            const int o = lane & 31, k = 64 * w + 8 * kb + 4 * (lane >> 5) + jj;
                            po[(((size_t)w * 8 + kb) * 64 + lane) * 4 + jj] = (o < O) ? w_out_h[(size_t)o * H + k] : 0.0f;
                        }
            HIP_TRY(hipMemcpy(m->ffp_wbo, po.data(), po.size() * sizeof(float), hipMemcpyHostToDevice));
        }
        if (!pre_only) {
            HIP_TRY(hipMemcpy(m->w_out, cur, (size_t)O * H * sizeof(float), hipMemcpyHostToDevice));
            cur += (size_t)O * H;
            HIP_TRY(hipMemcpy(m->b_out, cur, O * sizeof(float), hipMemcpyHostToDevice));
            m->has_weights = true;
            return APE_OK;
        }
    }
    for (int l = 0; l < L; ++l) {
        const int in_l = (l == 0) ? m->lstm_in : H;
        const int KXl = (l == 0) ? m->KX : H;
        const int Q = (KXl + H) / 16;
        const float* w_ih = cur;  cur += (size_t)4 * H * in_l;
        const float* w_hh = cur;  cur += (size_t)4 * H * H;
        const float* b_ih = cur;  cur += 4 * H;
        const float* b_hh = cur;  cur += 4 * H;
        std::vector<float> packed((size_t)4 * Q * NT * 64 * 4);
        for (int w = 0; w < 4; ++w)
            for (int q = 0; q < Q; ++q)
                for (int n = 0; n < NT; ++n) {
                    const int gate = n / UB, u = n % UB;
                    for (int lane = 0; lane < 64; ++lane) {
                        const int row = gate * H + w * (H / 4) + u * 16 + (lane & 15);
                        float* dst = &packed[((((size_t)w * Q + q) * NT + n) * 64 + lane) * 4];
                        for (int j = 0; j < 4; ++j) {
                            const int k = 16 * q + 4 * (lane >> 4) + j;
                            float v;
                            if (k < KXl) v = (k < in_l) ? w_ih[(size_t)row * in_l + k] : 0.0f;
                            else v = w_hh[(size_t)row * H + (k - KXl)];
                            dst[j] = v;
                        }
                    }
                }
        if (m->cluster_ok) {
            // cluster kernel: [member GH][wave 4][i/4][lane][i%4] with i = 4q + j; lane = (g << 4) | (u << 2) | gate holds
            // Wcat[gate*H + member*16 + wave*4 + u][16q + 4g + j] -- the register file of that wave
            const int GH = H / 16, NW = (KXl + H) / 4;
            std::vector<float> pc((size_t)GH * 4 * NW * 64);
            for (int mem = 0; mem < GH; ++mem)
                for (int w = 0; w < 4; ++w)
                    for (int i = 0; i < NW; ++i)
                        for (int lane = 0; lane < 64; ++lane) {
                            const int c = lane & 15, g = lane >> 4, u = c >> 2, gate = c & 3;   // tile row = unit*4 + gate
                            const int row = gate * H + mem * 16 + w * 4 + u;
                            const int k = 16 * (i / 4) + 4 * g + (i % 4);
                            float v;
                            if (k < KXl) v = (k < in_l) ? w_ih[(size_t)row * in_l + k] : 0.0f;
                            else v = w_hh[(size_t)row * H + (k - KXl)];
                            // register i of that lane; stored [k-quad i/4][lane][i%4] so a lane fetches 4 registers per load
                            pc[((((size_t)(mem * 4 + w) * (NW / 4)) + i / 4) * 64 + lane) * 4 + (i % 4)] = v;
                        }
            HIP_TRY(hipMemcpy(m->wcl[l], pc.data(), pc.size() * sizeof(float), hipMemcpyHostToDevice));
            if (m->small_uw == 2) {
                // latency kernel, H/8 members: a wave owns 2 units = 8 columns, 8 k-groups; lane = (g << 3) | (u << 2) | gate holds
                // Wcat[gate*H + (member*4 + wave)*2 + u][32q + 4g + j] in register 4q + j; same [i/4][lane][i%4] storage
                const int GS = H / 8, NWS = (KXl + H) / 8;
                std::vector<float> ps((size_t)GS * 4 * NWS * 64);
                for (int mem = 0; mem < GS; ++mem)
                    for (int w = 0; w < 4; ++w)
                        for (int i = 0; i < NWS; ++i)
                            for (int lane = 0; lane < 64; ++lane) {
                                const int c = lane & 7, g = lane >> 3, u = c >> 2, gate = c & 3;
                                const int row = gate * H + (mem * 4 + w) * 2 + u;
                                const int k = 32 * (i / 4) + 4 * g + (i % 4);
                                float v;
                                if (k < KXl) v = (k < in_l) ? w_ih[(size_t)row * in_l + k] : 0.0f;
                                else v = w_hh[(size_t)row * H + (k - KXl)];
                                ps[((((size_t)(mem * 4 + w) * (NWS / 4)) + i / 4) * 64 + lane) * 4 + (i % 4)] = v;
                            }
                HIP_TRY(hipMemcpy(m->wcls[l], ps.data(), ps.size() * sizeof(float), hipMemcpyHostToDevice));
            }
            if (m->mcs_ok && l >= 1) {
                // Monte-Carlo latency kernel, layers above layer 0 (v_mfma_f32_4x4x1_16b_f32): H/8 members x 4 waves, wave w = K quarter w of
                // [W_ih | W_hh]; lane = (block b = lane >> 2: unit b & 7 of the member, k slice ks = b >> 3; gate = lane & 3) holds
                // Wcat[gate*H + member*8 + (b & 7)][w*H/2 + 8 (i/4) + 4 ks + (i%4)] in register i; stored [member][wave][i/4][lane][i%4]
                const int GM = H / 8, NI4 = H / 4;
                std::vector<float> pm((size_t)GM * 4 * NI4 * 64);
                for (int mem = 0; mem < GM; ++mem)
                    for (int w = 0; w < 4; ++w)
                        for (int i = 0; i < NI4; ++i)
                            for (int lane = 0; lane < 64; ++lane) {
                                const int gate = lane & 3, ub = (lane >> 2) & 7, ks = lane >> 5;
                                const int row = gate * H + mem * 8 + ub;
                                const int k = w * (H / 2) + 8 * (i / 4) + 4 * ks + (i % 4);
                                pm[((((size_t)(mem * 4 + w) * (NI4 / 4)) + i / 4) * 64 + lane) * 4 + (i % 4)] =
                                    k < H ? w_ih[(size_t)row * H + k] : w_hh[(size_t)row * H + (k - H)];
                            }
                HIP_TRY(hipMemcpy(m->wmc[l], pm.data(), pm.size() * sizeof(float), hipMemcpyHostToDevice));
                // ... and the 16 x 16 x 4 image of its 16-row form: wave = (column tile ct = w & 1, K half kh = w >> 1); lane = (g << 4) | (u << 2) |
                // gate holds Wcat[gate*H + member*8 + ct*4 + u][kh*H + 16q + 4g + j] in register 4q + j; stored [member][wave][q][lane][4]
                const int NQm = H / 16;
                for (int mem = 0; mem < GM; ++mem)
                    for (int w = 0; w < 4; ++w)
                        for (int q = 0; q < NQm; ++q)
                            for (int lane = 0; lane < 64; ++lane) {
                                const int c = lane & 15, g = lane >> 4, u = c >> 2, gate = c & 3, ct = w & 1, kh = w >> 1;
                                const int row = gate * H + mem * 8 + ct * 4 + u;
                                for (int j = 0; j < 4; ++j) {
                                    const int k = 16 * q + 4 * g + j;
                                    pm[((((size_t)(mem * 4 + w) * NQm) + q) * 64 + lane) * 4 + j] =
                                        kh == 0 ? w_ih[(size_t)row * H + k] : w_hh[(size_t)row * H + k];
                                }
                            }
                HIP_TRY(hipMemcpy(m->wmc16[l], pm.data(), pm.size() * sizeof(float), hipMemcpyHostToDevice));
            }
            // fp16 variant: [member][wave][32-deep k-block q][lane][8]: lane holds Wcat[row][32q + 8g + j] as binary16
            const int NB = (KXl + H) / 32;
            std::vector<_Float16> ph((size_t)GH * 4 * NB * 64 * 8);
            for (int mem = 0; mem < GH; ++mem)
                for (int w = 0; w < 4; ++w)
                    for (int q = 0; q < NB; ++q)
                        for (int lane = 0; lane < 64; ++lane) {
                            const int c = lane & 15, g = lane >> 4, u = c >> 2, gate = c & 3;
                            const int row = gate * H + mem * 16 + w * 4 + u;
                            for (int j = 0; j < 8; ++j) {
                                const int k = 32 * q + 8 * g + j;
                                float v;
                                if (k < KXl) v = (k < in_l) ? w_ih[(size_t)row * in_l + k] : 0.0f;
                                else v = w_hh[(size_t)row * H + (k - KXl)];
                                ph[((((size_t)(mem * 4 + w) * NB) + q) * 64 + lane) * 8 + j] = (_Float16)v;
                            }
                        }
            if (m->wcl16[l]) HIP_TRY(hipMemcpy(m->wcl16[l], ph.data(), ph.size() * sizeof(_Float16), hipMemcpyHostToDevice));
        }
        if (m->wcl32[l] != nullptr) {
            // second-generation f32 cluster kernel (v_mfma_f32_32x32x2_f32, weights = A operand): 8 members x 4 waves, a wave owns
            // 8 units = 32 columns ordered gate * 8 + unit.  [member][wave][i / 4][lane][i % 4] with register i = 4 kb + j of
            // lane (column m = lane & 31, half hh = lane >> 5) = Wcat[gate(m) * H + member*32 + wave*8 + (m & 7)][8 kb + 4 hh + j]
            const int NW = (KXl + H) / 2;                   // registers per lane: 32 columns x K / 64 lanes
            std::vector<float> pc((size_t)8 * 4 * NW * 64);
            for (int mem = 0; mem < 8; ++mem)
                for (int w = 0; w < 4; ++w)
                    for (int i = 0; i < NW; ++i)
                        for (int lane = 0; lane < 64; ++lane) {
                            const int mcol = lane & 31, hh = lane >> 5;
                            const int row = (mcol >> 3) * H + mem * 32 + w * 8 + (mcol & 7);
                            const int k = 8 * (i / 4) + 4 * hh + (i % 4);
                            float v;
                            if (k < KXl) v = (k < in_l) ? w_ih[(size_t)row * in_l + k] : 0.0f;
                            else v = w_hh[(size_t)row * H + (k - KXl)];
                            pc[((((size_t)(mem * 4 + w) * (NW / 4)) + i / 4) * 64 + lane) * 4 + (i % 4)] = v;
                        }
            HIP_TRY(hipMemcpy(m->wcl32[l], pc.data(), pc.size() * sizeof(float), hipMemcpyHostToDevice));
        }
        if (m->up128_ok && l >= 1) {
            // lstm_upper128.hip: four members x 4 waves, a wave owns 8 units = 32 columns ordered gate * 8 + unit (v_mfma_f32_32x32x2_f32,
            // weights = A operand); register i = 4 kb + j of lane (column mcol = lane & 31, half hh) = Wcat[gate(mcol) * H + member*32 +
            // wave*8 + (mcol & 7)][8 kb + 4 hh + j]; [member][wave][i / 4][lane][i % 4]
            const int NW = (KXl + H) / 2;
            std::vector<float> pc((size_t)4 * 4 * NW * 64);
            for (int mem = 0; mem < 4; ++mem)
                for (int w = 0; w < 4; ++w)
                    for (int i = 0; i < NW; ++i)
                        for (int lane = 0; lane < 64; ++lane) {
                            const int mcol = lane & 31, hh = lane >> 5;
                            const int row = (mcol >> 3) * H + mem * 32 + w * 8 + (mcol & 7);
                            const int k = 8 * (i / 4) + 4 * hh + (i % 4);
                            pc[((((size_t)(mem * 4 + w) * (NW / 4)) + i / 4) * 64 + lane) * 4 + (i % 4)] =
                                k < KXl ? w_ih[(size_t)row * in_l + k] : w_hh[(size_t)row * H + (k - KXl)];
                        }
            HIP_TRY(hipMemcpy(m->wup128[l], pc.data(), pc.size() * sizeof(float), hipMemcpyHostToDevice));
        }
        std::vector<float> bsum(4 * H);
        for (int i = 0; i < 4 * H; ++i) bsum[i] = b_ih[i] + b_hh[i];
        HIP_TRY(hipMemcpy(m->wpack[l], packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(m->bias[l], bsum.data(), bsum.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    HIP_TRY(hipMemcpy(m->w_out, cur, (size_t)O * H * sizeof(float), hipMemcpyHostToDevice));
    cur += (size_t)O * H;
    HIP_TRY(hipMemcpy(m->b_out, cur, O * sizeof(float), hipMemcpyHostToDevice));
    m->has_weights = true;
    return APE_OK;
}

int ape_model_set_norm_stats(ape_model_t* m, const double* xx_m, const double* xx_s, const double* yy_m,
                             const double* yy_s) {
    if (!m || !xx_m || !xx_s || !yy_m || !yy_s) return fail(APE_ERR_INVALID_ARG, "set_norm_stats: NULL argument");
    const int I = m->dims.input_size, O = m->dims.output_size;
    std::vector<double> h(3 * I + 2 * O);
    for (int i = 0; i < I; ++i) h[2 * I + 2 * O + i] = 1.0 / xx_s[i];     // reciprocal for the in-kernel z-score
    memcpy(&h[0], xx_m, I * sizeof(double));
    memcpy(&h[I], xx_s, I * sizeof(double));
    memcpy(&h[2 * I], yy_m, O * sizeof(double));
    memcpy(&h[2 * I + O], yy_s, O * sizeof(double));
    HIP_TRY(hipSetDevice(m->dims.device));
    HIP_TRY(hipMemcpy(m->stats, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice));
    m->has_stats = true;
    return APE_OK;
}

int ape_model_set_body(ape_model_t* m, const double body9[9]) {
    if (!m || !body9) return fail(APE_ERR_INVALID_ARG, "set_body: NULL argument");
    memcpy(m->body, body9, 9 * sizeof(double));
    return APE_OK;
}

// the post-filter ape_infer wants behind the regressor: the latency kernel runs it in its own launch (fk_done = true), every other
// kernel leaves it to the caller
struct FkTail {
    void* est;
    int est_dtype;
    bool denormalize;
    bool done;
};

// Monte-Carlo latency kernel (lstm_mc_small.hip): n_streams windows (stream s at x + s * x_stream_stride) x n_mc dropout samples each,
// rows dealt over the 8 XCD clusters; y [n_streams * n_mc, O].  The caller has checked mc_small_fits().
static bool mc_small_fits(const ape_model* m, int n_streams, int n_mc) {
    if (!m->mcs_ok || !m->c32_on || m->kernel_choice != APE_KERNEL_AUTO || m->precision != APE_PRECISION_F32 || m->replaying) return false;
    if (n_streams < 1 || n_streams > 8 || n_mc < 1) return false;
    const int cps = 8 / n_streams;
    return (n_mc + cps - 1) / cps <= 16;
}
// what a host frame adds to the regressor's launch: the raw rows of the frame (feature builder workgroups)
struct McFrameParts {
    const float* raw_rows = nullptr;
    int raw_width = 0, raw_kind = 0, raw_big_endian = 0, cold = 0;
    float* ring_out = nullptr;
    size_t ring_stream_stride = 0, ring_rep_stride = 0;
    int ring_rep = 0;
};
// can the feature builder's workgroups become resident beside the clusters?  A cluster member takes a CU's worth of LDS only in the
// 16-row form; with 8-row clusters two workgroups share a CU, and the 3 x 128 model leaves half the CUs free anyway
static bool mc_small_builder_fits(const ape_model* m, int n_streams, int n_mc, int T) {
    const int cps = 8 / n_streams;
    // (8-row clusters: 39 KB of B tiles and sums + 0.5 KB of mask bits per step + 10.5 KB static per member -- two workgroups per CU up to T = 32)
    return m->dims.hidden_size == 128 || ((n_mc + cps - 1) / cps <= 8 && T <= 32);
}
static int mc_small_launch(ape_model_t* m, const float* x, size_t x_stream_stride, int n_streams, int n_mc, int T, uint32_t flags,
                           const float* masks_dev, float dropout_p, uint64_t seed, float* y_dev, void* stream, int x_ring,
                           const McFrameParts* fr = nullptr) {
    McSmallParams q{};
    const int L = m->dims.num_layers, I = m->dims.input_size, O = m->dims.output_size;
    q.x = x; q.x_stream_stride = x_stream_stride; q.y = y_dev;
    q.w0 = m->wcls[0];
    for (int l = 0; l < L; ++l) { q.w[l] = m->wmc[l]; q.w16[l] = m->wmc16[l]; q.bias[l] = m->bias[l]; }
    q.w_out = m->w_out; q.b_out = m->b_out;
    q.xx_m = m->stats; q.xx_s = m->stats + I; q.xx_r = m->stats + 2 * I + 2 * O;
    q.gx = m->gxm + 256; q.gx_cluster_bytes = (unsigned)m->gxm_cluster_bytes;
    q.seq = reinterpret_cast<unsigned*>(m->gxm);
    q.status = m->xflags + m->xflag_bytes / sizeof(unsigned); q.done = q.status - 3;
    q.xcc_slots = m->xcc_slots;
    q.masks = masks_dev;
    q.rows = n_streams * n_mc; q.n_mc = n_mc; q.n_streams = n_streams;
    q.cps = 8 / n_streams; q.R = (n_mc + q.cps - 1) / q.cps;
    q.T = T; q.I = I; q.O = O; q.x_ring = x_ring;
    q.flags = flags & (APE_FLAG_NORMALIZE_INPUT | APE_FLAG_DROPOUT_MASKS | APE_FLAG_DROPOUT_PHILOX | APE_FLAG_ANY_PLACEMENT);
    q.dropout_p = (flags & APE_FLAG_DROPOUT_PHILOX) ? dropout_p : 0.0f; q.seed = seed;
    q.dbg_wg = m->dbg_wg;
    q.xg = m->gxm + 256 + 8 * m->gxm_cluster_bytes;
    if (fr) {
        q.raw_rows = fr->raw_rows; q.raw_width = fr->raw_width; q.raw_kind = fr->raw_kind; q.raw_big_endian = fr->raw_big_endian;
        q.cold = fr->cold; q.ring_out = fr->ring_out; q.ring_stream_stride = fr->ring_stream_stride; q.ring_rep_stride = fr->ring_rep_stride;
        q.ring_rep = fr->ring_rep;
    }
    m->last_kernel = "ape_lstm_mc_small";
    hipError_t e = ape_launch_lstm_mc_small(m->dims.hidden_size, L, m->KX, q, (hipStream_t)stream);
    if (e != hipSuccess) return fail(APE_ERR_HIP, "Monte-Carlo latency kernel launch failed: %s", hipGetErrorString(e));
    return APE_OK;
}

// x_ring: time step t of every window lives in slot (t + x_ring) mod T (0 = the linear layout of the public entry)
static int lstm_forward_impl(ape_model_t* m, const float* x_dev, int32_t B, int32_t T, uint32_t flags,
                             const float* masks_dev, float dropout_p, uint64_t seed, float* y_dev, void* stream,
                             int x_ring, const float* h0_dev = nullptr, const float* c0_dev = nullptr, FkTail* fk = nullptr) {
    if (!m || !x_dev || !y_dev) return fail(APE_ERR_INVALID_ARG, "lstm_forward: NULL argument");
    if (B < 1 || T < 1) return fail(APE_ERR_INVALID_ARG, "lstm_forward: B=%d T=%d must be >= 1", B, T);
    flags &= ~(uint32_t)APE_FLAG_XCD_CLASSES;            // (the launcher's own bit)
#if !defined(APE_ABLATE) && !defined(APE_CLUSTER_STAMPS)
    // the product library knows the documented flags and the four exchange-form selectors of include/ape_hip.h; the timing-only
    // ablation bits (results are garbage) exist in the diagnostic builds alone
    if (flags & ~(uint32_t)(APE_FLAG_NORMALIZE_INPUT | APE_FLAG_ALL_STEPS | APE_FLAG_DROPOUT_MASKS | APE_FLAG_DROPOUT_PHILOX | APE_FLAG_BROADCAST_X |
                            APE_FLAG_ANY_PLACEMENT | APE_FLAG_IN_XCD_PLAIN | APE_FLAG_NO_XCD_CLASSES | APE_FLAG_ALT_FORM))
        return fail(APE_ERR_INVALID_ARG, "lstm_forward: unknown flag bits 0x%x", flags);
#endif
    if (!m->has_weights) return fail(APE_ERR_NOT_READY, "lstm_forward: weights not loaded");
    if ((flags & APE_FLAG_NORMALIZE_INPUT) && !m->has_stats)
        return fail(APE_ERR_NOT_READY, "lstm_forward: NORMALIZE_INPUT without norm stats");
    if ((flags & APE_FLAG_DROPOUT_MASKS) && (flags & APE_FLAG_DROPOUT_PHILOX))
        return fail(APE_ERR_INVALID_ARG, "lstm_forward: choose one dropout mode");
    if ((flags & APE_FLAG_DROPOUT_MASKS) && m->dims.num_layers > 1 && !masks_dev)
        return fail(APE_ERR_INVALID_ARG, "lstm_forward: DROPOUT_MASKS without masks");
    if ((flags & APE_FLAG_DROPOUT_PHILOX) && !(dropout_p >= 0.0f && dropout_p < 1.0f))
        return fail(APE_ERR_INVALID_ARG, "lstm_forward: dropout_p %f outside [0,1)", dropout_p);
    const int H = m->dims.hidden_size, L = m->dims.num_layers;
    const bool drop = (flags & (APE_FLAG_DROPOUT_MASKS | APE_FLAG_DROPOUT_PHILOX)) != 0;
    const bool have_hs = h0_dev != nullptr || c0_dev != nullptr;
    if (have_hs && (!h0_dev || !c0_dev)) return fail(APE_ERR_INVALID_ARG, "lstm_forward: h0 and c0 come together");
    if (have_hs && m->dims.model_kind == APE_MODEL_FF)
        return fail(APE_ERR_INVALID_ARG, "lstm_forward: the MLP regressor has no recurrent state");
    if (m->dims.model_kind == APE_MODEL_FF) {
        if (flags & APE_FLAG_BROADCAST_X) return fail(APE_ERR_UNSUPPORTED, "lstm_forward: BROADCAST_X is an LSTM-path flag");
        if ((flags & APE_FLAG_DROPOUT_MASKS) && !masks_dev)
            return fail(APE_ERR_INVALID_ARG, "lstm_forward: DROPOUT_MASKS without masks");
        MlpParams q{};
        const bool all = (flags & APE_FLAG_ALL_STEPS) != 0;
        q.x = x_dev; q.y = y_dev;
        for (int j = 0; j <= L; ++j) { q.wpack[j] = m->ff_wpack[j]; q.bias[j] = m->ff_bias[j]; }
        q.w_out = m->w_out; q.b_out = m->b_out;
        q.xx_m = m->stats; q.xx_s = m->stats + m->dims.input_size;
        q.mask = masks_dev;
        q.row_stride = all ? (size_t)m->dims.input_size : (size_t)T * m->dims.input_size;
        q.row_offset = all ? 0 : (size_t)(T - 1) * m->dims.input_size;
        q.N = all ? B * T : B;
        q.I = m->dims.input_size; q.O = m->dims.output_size; q.KX = m->KX; q.n_hidden = L;
        q.flags = flags; q.dropout_p = dropout_p; q.seed = seed;
        q.neg_slope = 0.01f;                      // torch's leaky_relu default (nn_models.py:347,349)
        // chip-filling eval batches: the weight-stationary two-stage pipeline (mlp_pipe.hip)
        // (x and y each behind one 32-bit buffer descriptor whose span stays UNDER 2 GiB: the kernel's out-of-range
        //  sentinel for padded columns and spare lanes is the offset 0x80000000, which must lie outside num_records)
        const bool one_descriptor = ((size_t)(q.N - 1) * q.row_stride + q.row_offset + q.I) * sizeof(float) < 0x80000000ull &&
                                    (size_t)q.N * q.O * sizeof(float) < 0x80000000ull;
        if (m->ffp_ok && m->ffp_on && !m->replaying && !drop && q.hidden_out == nullptr && q.mask == nullptr && one_descriptor && q.N >= 64 * m->n_cus) {
            hipError_t e2 = ape_launch_mlp_pipe(q, m->ffp_wa0, m->ffp_wa1, m->ffp_wb2, m->ffp_wbo, m->ffp_ring, m->ffp_ring_bytes, m->ffp_ctl,
                                               m->n_cus, (hipStream_t)stream);
            if (e2 != hipSuccess) return fail(APE_ERR_HIP, "mlp pipeline launch failed: %s", hipGetErrorString(e2));
            return APE_OK;
        }
        hipError_t e = ape_launch_mlp_tile16(H, q, (hipStream_t)stream, m->n_cus);
        if (e != hipSuccess) return fail(APE_ERR_HIP, "mlp kernel launch failed: %s", hipGetErrorString(e));
        return APE_OK;
    }
    const float* lstm_x = x_dev;
    if (m->dims.model_kind == APE_MODEL_IMUPOSE) {
        // relu(input_layer(x)) for every row of every window (nn_models.py:242), then the LSTM reads those 256 columns
        if (drop) return fail(APE_ERR_UNSUPPORTED, "lstm_forward: ImuPoseLSTM has no Monte-Carlo dropout mode "
                              "(its monte_carlo_predictions is the plain forward, nn_models.py:246-251)");
        const size_t rows = (size_t)((flags & APE_FLAG_BROADCAST_X) ? 1 : B) * T;
        if (rows > m->z_cap) {
            hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
            (void)hipStreamIsCapturing((hipStream_t)stream, &st);
            if (st != hipStreamCaptureStatusNone)
                return fail(APE_ERR_CAPACITY, "lstm_forward: %zu input-layer rows exceed the workspace during stream capture", rows);
            if (m->z_ws) { HIP_TRY(hipFree(m->z_ws)); m->z_ws = nullptr; m->z_cap = 0; }
            HIP_TRY(hipMalloc((void**)&m->z_ws, rows * H * sizeof(float)));
            m->z_cap = rows;
        }
        MlpParams q{};
        q.x = x_dev; q.y = nullptr; q.hidden_out = m->z_ws;
        q.wpack[0] = m->ff_wpack[0]; q.bias[0] = m->ff_bias[0];
        q.xx_m = m->stats; q.xx_s = m->stats + m->dims.input_size;
        q.row_stride = (size_t)m->dims.input_size; q.row_offset = 0;
        q.N = (int)rows; q.I = m->dims.input_size; q.O = m->dims.output_size; q.KX = m->KXpre; q.n_hidden = 0;
        q.flags = flags & APE_FLAG_NORMALIZE_INPUT; q.neg_slope = 0.0f;
        hipError_t e = ape_launch_mlp_tile16(H, q, (hipStream_t)stream, m->n_cus);
        if (e != hipSuccess) return fail(APE_ERR_HIP, "input-layer kernel launch failed: %s", hipGetErrorString(e));
        lstm_x = m->z_ws;
        flags &= ~(uint32_t)APE_FLAG_NORMALIZE_INPUT;        // done in front of the input layer
        // Above 512 windows (where the first-generation kernel needs a second launch: 1480 us for 513 .. 1024 windows x 64 steps against
        // 1020-1060 here) the LSTM runs one layer per launch on the persistent clusters of lstm_upper32.hip: layer 0 in the SEQ form
        // with the wide input, layer 1 reading its sequence as is -- K = 512 per layer is the clusters' whole register image; the
        // first-generation kernel's 16-member clusters re-read nothing either but spend 16 CUs on 32 rows.
        // Chunks of 4096 windows bound the workspaces.
        if (m->split32_ok && B > 512 && m->kernel_choice == APE_KERNEL_AUTO && m->c32_on && m->precision == APE_PRECISION_F32 &&
            !m->replaying && !have_hs && !(flags & (APE_FLAG_ALL_STEPS | APE_FLAG_BROADCAST_X)) && x_ring == 0 && T >= 1 &&
            (size_t)128 * T * 32768 < ((size_t)1 << 32)) {
            const int chunk = 4096;
            const int tiles_max = ((B < chunk ? B : chunk) + 31) / 32;
            if ((size_t)tiles_max > m->split_tiles_cap || (size_t)T > m->split_steps_cap) {
                hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
                (void)hipStreamIsCapturing((hipStream_t)stream, &st);
                if (st != hipStreamCaptureStatusNone)
                    return fail(APE_ERR_CAPACITY, "lstm_forward: the layer workspaces (%d tiles x %d steps) cannot grow during stream capture", tiles_max, T);
                HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
                if (m->zfrag_ws) { HIP_TRY(hipFree(m->zfrag_ws)); m->zfrag_ws = nullptr; }
                if (m->hfrag_ws) { HIP_TRY(hipFree(m->hfrag_ws)); m->hfrag_ws = nullptr; }
                if (m->ypart_ws) { HIP_TRY(hipFree(m->ypart_ws)); m->ypart_ws = nullptr; }
                // (both dimensions only ever grow: a caller alternating between many short and few long windows does not reallocate)
                const size_t tc = (size_t)tiles_max > m->split_tiles_cap ? (size_t)tiles_max : m->split_tiles_cap;
                const size_t sc = (size_t)T > m->split_steps_cap ? (size_t)T : m->split_steps_cap;
                m->split_tiles_cap = m->split_steps_cap = 0;
                HIP_TRY(hipMalloc((void**)&m->zfrag_ws, tc * sc * 32768));
                HIP_TRY(hipMalloc((void**)&m->hfrag_ws, tc * sc * 32768));
                HIP_TRY(hipMalloc((void**)&m->ypart_ws, ape_upper32_ypart_bytes((int)tc * 32)));
                m->split_tiles_cap = tc; m->split_steps_cap = sc;
            }
            for (int b0 = 0; b0 < B; b0 += chunk) {
                const int nb = (B - b0 < chunk) ? B - b0 : chunk;
                const int tiles = (nb + 31) / 32;
                UpperParams u0{};
                u0.xfrag = m->zfrag_ws; u0.xfrag_bytes = (size_t)tiles * T * 32768;
                u0.w = m->wcl32[0]; u0.bias = m->bias[0]; u0.w_out = m->w_out;
                u0.hx = m->hx; u0.hx_bytes = m->hx_bytes;
                u0.xflags = m->xflags; u0.status = m->xflags + m->xflag_bytes / sizeof(unsigned); u0.done = u0.status - 3;
                u0.xcc_slots = m->xcc_slots; u0.dbg_wg = m->dbg_wg;
                u0.hseq = m->hfrag_ws; u0.hseq_bytes = (size_t)tiles * T * 32768;
                u0.T = T; u0.O = m->dims.output_size; u0.n_tiles = tiles; u0.flags = flags & APE_FLAG_ALT_FORM;      // (selector: one-tile clusters on the blocking form, for A/B)
                UpperParams u1 = u0;
                u1.xfrag = m->hfrag_ws; u1.hseq = nullptr; u1.hseq_bytes = 0;
                u1.w = m->wcl32[1]; u1.bias = m->bias[1]; u1.ypart = m->ypart_ws;
                m->last_kernel = "ape_lstm_upper32";
                hipError_t e2 = ape_launch_lstm_split32(m->z_ws + (size_t)b0 * T * H, nb, u0, u1, m->b_out,
                                                        y_dev + (size_t)b0 * m->dims.output_size, f16v2_capacity(m->n_cus), (hipStream_t)stream);
                if (e2 != hipSuccess) return fail(APE_ERR_HIP, "layer-split lstm launch failed: %s", hipGetErrorString(e2));
            }
            return APE_OK;
        }
    }
    if (ape_lstm_tile16_smem_bytes(H, L, m->KX, m->dims.output_size, drop) > 160 * 1024)
        return fail(APE_ERR_UNSUPPORTED, "lstm_forward: H=%d L=%d with dropout exceeds the 160 KiB LDS of a CU", H, L);

    // Two kernels serve an LSTM batch.  The weight-stationary cluster kernel fills the chip from one launch of
    // 1..1024 rows (512 with inter-layer dropout) and is the faster one per row on long windows; the batch-tile kernel
    // needs 4096 rows (256 workgroups x 16) to fill the chip, but then runs dropout at no extra cost and pays no
    // per-launch prologue / head, which decides short windows.  Under APE_KERNEL_AUTO the front of the batch goes to
    // the batch-tile kernel in whole 4096-row waves and the rest to the cluster kernel, by a cost model calibrated
    // on MI355X (tests/tools/time_big_batch.py).
    // (a re-issue by ape_model_recover runs on the batch-tile kernel, in exact float32 whatever the precision switch)
    const bool f16 = m->precision == APE_PRECISION_F16 && !m->replaying;
    // all-steps output: the cluster kernel also writes every step's top-layer output to a [B,T,H] workspace and the
    // head runs over those rows in a second, HBM-bound launch
    const bool all_steps = (flags & APE_FLAG_ALL_STEPS) != 0;
    const bool cdrop_c = drop && L > 1;
    const int rows_per_cluster_launch = cluster_rows_per_launch(m->n_cus, H, cdrop_c || m->wide_cluster);
    // injected masks are indexed over the whole batch, so such a call is served by ONE launch of one kernel
    const bool masks_fit = !(flags & APE_FLAG_DROPOUT_MASKS) || !cdrop_c || B <= rows_per_cluster_launch ||
                           m->kernel_choice == APE_KERNEL_CLUSTER;
    // a caller-given initial state (h0, c0) is served by the batch-tile kernel, which loads it at step 0
    bool use_cluster = m->cluster_ok && masks_fit && m->kernel_choice != APE_KERNEL_TILE16 && !(all_steps && f16) && !have_hs &&
                       !m->replaying;
    if (have_hs && f16) return fail(APE_ERR_UNSUPPORTED, "lstm_forward: the fp16 variant starts from the zero state only");
    if (f16 && (!m->cluster_ok || drop || (flags & APE_FLAG_ALL_STEPS)))
        return fail(APE_ERR_UNSUPPORTED, "lstm_forward: the fp16 variant covers last-step output without dropout on "
                    "the cluster-kernel shapes only");
    if (f16) use_cluster = true;
    if (m->kernel_choice == APE_KERNEL_CLUSTER && !use_cluster && !m->replaying)
        return fail(APE_ERR_UNSUPPORTED, "lstm_forward: the cluster kernel does not cover this model / these flags");
    // one window, n dropout samples (monte_carlo_predictions, nn_models.py:191-207) up to 128 rows: the Monte-Carlo latency kernel
    if (use_cluster && cdrop_c && (flags & APE_FLAG_BROADCAST_X) && !all_steps && !f16 && B <= 128 && T <= 64 && mc_small_fits(m, 1, B))
        return mc_small_launch(m, x_dev, 0, 1, B, T, flags, masks_dev, dropout_p, seed, y_dev, stream, x_ring);
    int n16 = use_cluster ? 0 : B;               // leading rows that go to the batch-tile kernel
    if (use_cluster && m->kernel_choice == APE_KERNEL_AUTO && !f16 && !all_steps && !(flags & APE_FLAG_DROPOUT_MASKS) && B > 4) {
        const int w = auto_tile16_waves(&m->dims, m->n_cus, B, T, cdrop_c, drop ? 0 : gen2_of(m), m->wide_cluster);
        const long long front = (long long)tile16_wave_rows(m->n_cus) * w;
        n16 = (front < B) ? (int)front : B;
    }
    if (n16 > 0) {
        LstmParams p{};
        p.x = lstm_x;
        p.y = y_dev;
        for (int l = 0; l < L; ++l) { p.wpack[l] = m->wpack[l]; p.bias[l] = m->bias[l]; }
        p.w_out = m->w_out;
        p.b_out = m->b_out;
        p.xx_m = m->stats;
        p.xx_s = m->stats + m->dims.input_size;
        p.masks = masks_dev;
        p.B = n16; p.T = T; p.I = m->lstm_in; p.O = m->dims.output_size; p.KX = m->KX;
        p.x_ring = x_ring;
        p.flags = flags;
        p.dropout_p = dropout_p;
        p.seed = seed;
        p.h0 = h0_dev; p.c0 = c0_dev; p.hs_rows = B;
        m->last_kernel = "ape_lstm_tile16";
        hipError_t e = ape_launch_lstm_tile16(H, L, p, (hipStream_t)stream);
        if (e != hipSuccess) return fail(APE_ERR_HIP, "lstm kernel launch failed: %s", hipGetErrorString(e));
        if (n16 == B) return APE_OK;
    }
    if (use_cluster) {
        // smallest row tile count that still fits the batch on the chip: more clusters = more CUs busy
        const bool cdrop = drop && L > 1;
        const int nmt = cluster_nmt(m->n_cus, H, B - n16, cdrop || m->wide_cluster);      // (wide: at most two row tiles, like dropout)
        const int rows_per_launch = 16 * nmt * cluster_capacity(m->n_cus, H);
        const bool small = !f16 && !cdrop && !all_steps && B <= 4 && T + L <= 4096 && m->small_batch_path && !m->wide_cluster;   // latency path: VALU GEMV, one exchange per phase
        const int small_uw = (flags & APE_FLAG_ALT_FORM) ? 4 : m->small_uw;
        const int rest_plan = (f16 || small) ? PLAN_NONE : rest_kernel(B - n16, T, (!cdrop && !drop && !all_steps) ? gen2_of(m) : 0, m->n_cus);
        if (rest_plan == PLAN_C32) {
            // second-generation f32 kernel: 8-member clusters of 32 windows, 32x32x2 MFMA chain (lstm_cluster32.hip)
            const int rpl2 = 32 * f16v2_capacity(m->n_cus);
            for (int b0 = n16; b0 < B; b0 += rpl2) {
                const int nb = (B - b0 < rpl2) ? B - b0 : rpl2;
                ClusterParams c{};
                c.x = (flags & APE_FLAG_BROADCAST_X) ? x_dev : x_dev + (size_t)b0 * T * m->dims.input_size;
                c.y = y_dev + (size_t)b0 * m->dims.output_size;
                for (int l = 0; l < L; ++l) { c.wcl[l] = m->wcl32[l]; c.bias[l] = m->bias[l]; }
                c.w_out = m->w_out; c.b_out = m->b_out;
                c.xx_m = m->stats; c.xx_s = m->stats + m->dims.input_size;
                c.xx_r = m->stats + 2 * m->dims.input_size + 2 * m->dims.output_size;
                c.hx = m->hx; c.hx_bytes = m->hx_bytes;
                c.xflags = m->xflags; c.status = m->xflags + m->xflag_bytes / sizeof(unsigned);
                c.ticket = c.status - 4; c.done = c.status - 3;
                c.B = nb; c.T = T; c.I = m->dims.input_size; c.O = m->dims.output_size;
                c.flags = flags; c.x_ring = x_ring;
                c.xcc_slots = m->xcc_slots;
                c.dbg_wg = m->dbg_wg;
                m->last_kernel = "ape_lstm_cluster32";
                hipError_t e = ape_launch_lstm_cluster32(H, L, m->KX, (nb + 31) / 32, c, (hipStream_t)stream);
                if (e != hipSuccess) return fail(APE_ERR_HIP, "cluster32 lstm launch failed: %s", hipGetErrorString(e));
            }
            return APE_OK;
        }
        if (rest_plan == PLAN_C16) {
            // second-generation kernel of the 3 x 128 model: 8-member clusters of 32 windows (lstm_cluster16.hip)
            const int rpl2 = 32 * f16v2_capacity(m->n_cus);
            for (int b0 = n16; b0 < B; b0 += rpl2) {
                const int nb = (B - b0 < rpl2) ? B - b0 : rpl2;
                ClusterParams c{};
                c.x = (flags & APE_FLAG_BROADCAST_X) ? x_dev : x_dev + (size_t)b0 * T * m->dims.input_size;
                c.y = y_dev + (size_t)b0 * m->dims.output_size;
                for (int l = 0; l < L; ++l) { c.wcl[l] = m->wcl[l]; c.bias[l] = m->bias[l]; }
                c.w_out = m->w_out; c.b_out = m->b_out;
                c.xx_m = m->stats; c.xx_s = m->stats + m->dims.input_size;
                c.xx_r = m->stats + 2 * m->dims.input_size + 2 * m->dims.output_size;
                c.hx = m->hx; c.hx_bytes = m->hx_bytes;
                c.xflags = m->xflags; c.status = m->xflags + m->xflag_bytes / sizeof(unsigned);
                c.ticket = c.status - 4; c.done = c.status - 3;
                c.B = nb; c.T = T; c.I = m->dims.input_size; c.O = m->dims.output_size;
                c.flags = flags; c.x_ring = x_ring;
                c.xcc_slots = m->xcc_slots;
                c.dbg_wg = m->dbg_wg;
                m->last_kernel = "ape_lstm_cluster16";
                hipError_t e = ape_launch_lstm_cluster16(H, L, m->KX, nb, c, (hipStream_t)stream);
                if (e != hipSuccess) return fail(APE_ERR_HIP, "cluster16 lstm launch failed: %s", hipGetErrorString(e));
            }
            return APE_OK;
        }
        if (rest_plan == PLAN_LV16) {
            // short windows of the 3 x 128 model: level-synchronous 16-window clusters, two workgroups per CU (lstm_level16.hip)
            const int rpl3 = 32 * ape_level16_max_clusters(m->n_cus);
            for (int b0 = n16; b0 < B; b0 += rpl3) {
                const int nb = (B - b0 < rpl3) ? B - b0 : rpl3;
                ClusterParams c{};
                c.x = (flags & APE_FLAG_BROADCAST_X) ? x_dev : x_dev + (size_t)b0 * T * m->dims.input_size;
                c.y = y_dev + (size_t)b0 * m->dims.output_size;
                for (int l = 0; l < L; ++l) { c.wcl[l] = m->wcl[l]; c.bias[l] = m->bias[l]; }
                c.w_out = m->w_out; c.b_out = m->b_out;
                c.xx_m = m->stats; c.xx_s = m->stats + m->dims.input_size;
                c.xx_r = m->stats + 2 * m->dims.input_size + 2 * m->dims.output_size;
                c.hx = reinterpret_cast<float*>(m->gx16 + 256); c.hx_bytes = m->gx16_bytes; c.seq = reinterpret_cast<unsigned*>(m->gx16);
                c.xflags = m->xflags; c.status = m->xflags + m->xflag_bytes / sizeof(unsigned);
                c.ticket = c.status - 4; c.done = c.status - 3;
                c.B = nb; c.T = T; c.I = m->dims.input_size; c.O = m->dims.output_size;
                c.flags = flags & ~APE_FLAG_LV16_SINGLE; c.x_ring = x_ring;
                // up to 16 rows per cluster of the device: one row tile per cluster, so that the rows spread over every CU
                if (nb <= 16 * ape_level16_max_clusters(m->n_cus)) c.flags |= APE_FLAG_LV16_SINGLE;
                c.xcc_slots = m->xcc_slots;
                c.dbg_wg = m->dbg_wg;
                m->last_kernel = "ape_lstm_level16";
                hipError_t e = ape_launch_lstm_level16(H, L, m->KX, nb, c, (hipStream_t)stream);
                if (e != hipSuccess) return fail(APE_ERR_HIP, "level16 lstm launch failed: %s", hipGetErrorString(e));
            }
            return APE_OK;
        }
        if (f16 && m->f16_v2 && ape_cluster_f16v2_supported(H, L, m->KX) && f16v2_capacity(m->n_cus) > 0 && B > 256) {
            // second-generation fp16 kernel: 8-member clusters x 2 row sets of 16 that take turns (lstm_cluster_f16v2.hip)
            const int rpl2 = 32 * f16v2_capacity(m->n_cus);
            for (int b0 = 0; b0 < B; b0 += rpl2) {
                const int nb = (B - b0 < rpl2) ? B - b0 : rpl2;
                ClusterParams c{};
                c.x = (flags & APE_FLAG_BROADCAST_X) ? x_dev : x_dev + (size_t)b0 * T * m->dims.input_size;
                c.y = y_dev + (size_t)b0 * m->dims.output_size;
                for (int l = 0; l < L; ++l) { c.wcl[l] = reinterpret_cast<const float*>(m->wcl16[l]); c.bias[l] = m->bias[l]; }
                c.w_out = m->w_out; c.b_out = m->b_out;
                c.xx_m = m->stats; c.xx_s = m->stats + m->dims.input_size;
                c.xx_r = m->stats + 2 * m->dims.input_size + 2 * m->dims.output_size;
                c.hx = m->hx; c.hx_bytes = m->hx_bytes;
                c.xflags = m->xflags; c.status = m->xflags + m->xflag_bytes / sizeof(unsigned);
                c.ticket = c.status - 4; c.done = c.status - 3;
                c.B = nb; c.T = T; c.I = m->dims.input_size; c.O = m->dims.output_size;
                c.flags = flags; c.x_ring = x_ring;
                c.xcc_slots = m->xcc_slots;
                c.dbg_wg = m->dbg_wg;
                m->last_kernel = "ape_lstm_cluster_f16v2";
                // (APE_FLAG_ALT_FORM: the 16-unit-member form, two workgroups per CU -- needs 2 x 16 x clusters workgroups resident)
                const bool duo = (flags & APE_FLAG_ALT_FORM) != 0 && m->n_cus * 2 >= 16 * (((nb + 31) / 32 + 7) / 8 * 8);
                if (duo) m->last_kernel = "ape_lstm_cluster_f16v2<duo>";
                hipError_t e = ape_launch_lstm_cluster_f16v2(H, L, m->KX, (nb + 31) / 32, c, (hipStream_t)stream, duo);
                if (e != hipSuccess) return fail(APE_ERR_HIP, "fp16 cluster lstm launch failed: %s", hipGetErrorString(e));
            }
            return APE_OK;
        }
        if (all_steps) {
            const size_t rows = (size_t)B * T;
            if (rows > m->hseq_cap) {
                hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
                (void)hipStreamIsCapturing((hipStream_t)stream, &st);
                if (st != hipStreamCaptureStatusNone)
                    return fail(APE_ERR_CAPACITY, "lstm_forward: the all-steps workspace (%zu rows) cannot grow during stream capture", rows);
                if (m->hseq_ws) { HIP_TRY(hipFree(m->hseq_ws)); m->hseq_ws = nullptr; m->hseq_cap = 0; }
                HIP_TRY(hipMalloc((void**)&m->hseq_ws, rows * H * sizeof(float)));
                m->hseq_cap = rows;
            }
        }
        for (int b0 = n16; b0 < B; b0 += rows_per_launch) {
            const int nb = (B - b0 < rows_per_launch) ? B - b0 : rows_per_launch;
            ClusterParams c{};
            // (lstm_x / lstm_in: the LSTM's own input -- ImuPoseLSTM's is the 256-wide activation of its input layer)
            c.x = (flags & APE_FLAG_BROADCAST_X) ? lstm_x : lstm_x + (size_t)b0 * T * m->lstm_in;
            c.y = all_steps ? nullptr : y_dev + (size_t)b0 * m->dims.output_size;
            c.hseq = all_steps ? m->hseq_ws + (size_t)b0 * T * H : nullptr;
            for (int l = 0; l < L; ++l) {
                c.wcl[l] = f16 ? reinterpret_cast<const float*>(m->wcl16[l]) : (small && small_uw == 2) ? m->wcls[l] : m->wcl[l];
                c.bias[l] = m->bias[l];
            }
            c.w_out = m->w_out; c.b_out = m->b_out;
            c.xx_m = m->stats; c.xx_s = m->stats + m->dims.input_size;
            c.xx_r = m->stats + 2 * m->dims.input_size + 2 * m->dims.output_size;
            c.hx = m->hx; c.hx_bytes = m->hx_bytes;
            if (small) { c.hx = reinterpret_cast<float*>(m->hxs + 256); c.hx_bytes = m->hxs_bytes; c.seq = reinterpret_cast<unsigned*>(m->hxs); }
            c.xflags = m->xflags; c.status = m->xflags + m->xflag_bytes / sizeof(unsigned);
            c.ticket = m->xflags + m->xflag_bytes / sizeof(unsigned) - 4;
            c.done = c.ticket + 1;
            c.B = nb; c.T = T; c.I = m->lstm_in; c.O = m->dims.output_size;
            c.flags = flags & ~(uint32_t)APE_FLAG_ALL_STEPS;
            c.x_ring = x_ring;
            // injected masks are indexed [L-1, B, T, H] over the WHOLE batch: chunks need the full B stride,
            // so a masked call is served by one launch only (checked below)
            c.masks = masks_dev; c.dropout_p = dropout_p; c.seed = seed;
            c.dbg_wg = m->dbg_wg;
            c.xcc_slots = m->xcc_slots;
            if (small && fk != nullptr) {             // (B <= 4: one launch)
                c.fk_est = fk->est; c.fk_est_dtype = fk->est_dtype;
                if (fk->denormalize) {
                    c.fk_yy_m = m->stats + 2 * m->dims.input_size;
                    c.fk_yy_s = c.fk_yy_m + m->dims.output_size;
                }
                memcpy(c.fk_body, m->body, sizeof(c.fk_body));
                c.fk_layout = m->dims.target_layout; c.fk_W = layout_est_width(c.fk_layout);
                fk->done = true;
            }
            if ((flags & APE_FLAG_DROPOUT_MASKS) && b0 + nb < B && cdrop)
                return fail(APE_ERR_UNSUPPORTED, "lstm_forward: injected masks with B=%d exceed one cluster launch", B);
            c.row_base = b0;                          // Philox counters over the call's global rows, whatever the split
            int clusters = (nb + 16 * nmt - 1) / (16 * nmt);
            // f32 first-generation kernel: whole groups of 8 clusters (if the device holds them) form XCD-local clusters and hand
            // their slices over inside that XCD's L2 (lstm_cluster.hip, APE_FLAG_XCD_CLASSES); the extra clusters own no rows
            if (!small && !f16 && m->gen1_classes && !(flags & APE_FLAG_NO_XCD_CLASSES)) {   // (selector: A/B on one box)
                const int c8 = (clusters + 7) / 8 * 8;
                // (a launch of fewer than four clusters -- one stream's 25 Monte-Carlo rows -- stays as it is: there the rendezvous costs
                //  more than the shorter hops save, 44.7 vs 43.8 us)
                if (clusters >= 4 && c8 <= cluster_capacity(m->n_cus, H)) { clusters = c8; c.flags |= APE_FLAG_XCD_CLASSES; }
            }
            // no memset in the launch path: the kernel's last workgroup re-zeroes every polled word (self-cleaning)
            m->last_kernel = small ? "ape_lstm_cluster_small" : f16 ? "ape_lstm_cluster_f16" : "ape_lstm_cluster";
            hipError_t e = small ? ape_launch_lstm_cluster_small(H, L, m->KX, B == 1 ? 1 : (B == 2 ? 2 : 4), small_uw, c, (hipStream_t)stream)
                           : f16 ? ape_launch_lstm_cluster_f16(H, L, m->KX, nmt, clusters, c, (hipStream_t)stream)
                                 : ape_launch_lstm_cluster(H, L, m->KX, nmt, cdrop, clusters, c, (hipStream_t)stream);
            if (e != hipSuccess) return fail(APE_ERR_HIP, "cluster lstm launch failed: %s", hipGetErrorString(e));
        }
        if (all_steps) {
            hipError_t e = ape_launch_head_rows(m->hseq_ws, B * T, H, m->dims.output_size, m->w_out, m->b_out, y_dev,
                                                (hipStream_t)stream);
            if (e != hipSuccess) return fail(APE_ERR_HIP, "head kernel launch failed: %s", hipGetErrorString(e));
        }
        return APE_OK;
    }

    return fail(APE_ERR_UNSUPPORTED, "lstm_forward: no kernel covers this call");     // not reached
}

int ape_lstm_forward(ape_model_t* m, const float* x_dev, int32_t B, int32_t T, uint32_t flags,
                     const float* masks_dev, float dropout_p, uint64_t seed, float* y_dev, void* stream) {
    const int rc = lstm_forward_impl(m, x_dev, B, T, flags, masks_dev, dropout_p, seed, y_dev, stream, 0);
    if (rc == APE_OK) {
        ApeJournalEntry e{};
        e.kind = ApeJournalEntry::FORWARD; e.in0 = x_dev; e.in1 = masks_dev; e.out0 = y_dev; e.B = B; e.T = T; e.flags = flags;
        e.dropout_p = dropout_p; e.seed = seed; e.stream = stream;
        journal_add(m, e);
    }
    return rc;
}

int ape_lstm_forward_hs(ape_model_t* m, const float* x_dev, int32_t B, int32_t T, uint32_t flags,
                        const float* masks_dev, float dropout_p, uint64_t seed, const float* h0_dev,
                        const float* c0_dev, float* y_dev, void* stream) {
    // (an initial state is served by the batch-tile kernel, which cannot abort: journaled all the same, a later call may)
    const int rc = lstm_forward_impl(m, x_dev, B, T, flags, masks_dev, dropout_p, seed, y_dev, stream, 0, h0_dev, c0_dev);
    if (rc == APE_OK && masks_dev == nullptr) {
        ApeJournalEntry e{};
        e.kind = ApeJournalEntry::FORWARD_HS; e.in0 = x_dev; e.in1 = h0_dev; e.in2 = c0_dev; e.out0 = y_dev; e.B = B; e.T = T;
        e.flags = flags; e.dropout_p = dropout_p; e.seed = seed; e.stream = stream;
        journal_add(m, e);
    } else if (rc == APE_OK && m) {
        m->journal_overflow = true;          // (initial state AND injected masks: not representable in an entry)
    }
    return rc;
}

int ape_model_set_kernel(ape_model_t* m, int32_t choice) {
    if (!m) return fail(APE_ERR_INVALID_ARG, "set_kernel: NULL model");
    if (choice != APE_KERNEL_AUTO && choice != APE_KERNEL_TILE16 && choice != APE_KERNEL_CLUSTER && choice != APE_KERNEL_CLUSTER_GEN1 &&
        choice != APE_KERNEL_AUTO_GEN1)
        return fail(APE_ERR_INVALID_ARG, "set_kernel: unknown choice %d", choice);
    m->c32_on = choice != APE_KERNEL_CLUSTER_GEN1 && choice != APE_KERNEL_AUTO_GEN1;
    m->ffp_on = choice != APE_KERNEL_TILE16;           // (MLP regressor: TILE16 pins the tile kernel)
    if (choice == APE_KERNEL_CLUSTER_GEN1) choice = APE_KERNEL_CLUSTER;
    if (choice == APE_KERNEL_AUTO_GEN1) choice = APE_KERNEL_AUTO;
    if (choice == APE_KERNEL_CLUSTER && !m->cluster_ok)
        return fail(APE_ERR_UNSUPPORTED, "set_kernel: no cluster kernel for H=%d L=%d on a device with %d CUs (a cluster "
                    "needs %d)", m->dims.hidden_size, m->dims.num_layers, m->n_cus, m->dims.hidden_size / 16);
    m->kernel_choice = choice;
    m->small_batch_path = (choice == APE_KERNEL_AUTO);
    return APE_OK;
}

int ape_model_set_precision(ape_model_t* m, int32_t precision) {
    if (!m) return fail(APE_ERR_INVALID_ARG, "set_precision: NULL model");
    if (precision != APE_PRECISION_F32 && precision != APE_PRECISION_F16 && precision != APE_PRECISION_F16_GEN1)
        return fail(APE_ERR_INVALID_ARG, "set_precision: unknown precision %d", precision);
    m->f16_v2 = precision != APE_PRECISION_F16_GEN1;
    if (precision == APE_PRECISION_F16_GEN1) precision = APE_PRECISION_F16;
    if (precision == APE_PRECISION_F16 && (!m->cluster_ok || m->wide_cluster))
        return fail(APE_ERR_UNSUPPORTED, "set_precision: no fp16 kernel for H=%d L=%d", m->dims.hidden_size,
                    m->dims.num_layers);
    m->precision = precision;
    return APE_OK;
}

// the blocking part both ape_model_check and ape_model_recover share: 0 = no aborted launch since the last check; else the
// handle has been reset and g_err describes what happened
static int check_and_reset(ape_model_t* m);

int ape_model_check(ape_model_t* m) {
    if (!m) return fail(APE_ERR_INVALID_ARG, "check: NULL model");
    const int rc = check_and_reset(m);
    if (rc != APE_OK && rc != APE_ERR_INVALID_ARG) {
        m->stats_counts.aborted_checks += 1;
        m->stats_counts.lost_calls += (uint64_t)m->journal_n + (m->journal_overflow ? 1u : 0u);
    }
    journal_clear(m);
    return rc;
}

int ape_model_stats(const ape_model_t* m, ape_model_stats_t* out) {
    if (!m || !out) return fail(APE_ERR_INVALID_ARG, "stats: NULL argument");
    *out = m->stats_counts;
    return APE_OK;
}

static int replay_entry(ape_model_t* m, const ApeJournalEntry& e);

int ape_model_recover(ape_model_t* m) {
    if (!m) return fail(APE_ERR_INVALID_ARG, "recover: NULL model");
    const int rc = check_and_reset(m);
    if (rc == APE_OK) { journal_clear(m); return APE_OK; }
    if (rc == APE_ERR_INVALID_ARG) return rc;
    const std::string what = g_err;
    m->stats_counts.aborted_checks += 1;
    // can every pending call be issued again?  A bank step only while it is that bank's newest step with no row behind it.
    bool ok = !m->journal_overflow;
    for (int i = 0; i < m->journal_n && ok; ++i) {
        const ApeJournalEntry& e = m->journal[i];
        if (e.kind != ApeJournalEntry::STEP) continue;
        if (e.bank->frames != e.bank_frames || e.bank->steps != e.bank_steps + 1) ok = false;
        for (int j = i + 1; j < m->journal_n; ++j)
            if (m->journal[j].kind == ApeJournalEntry::STEP && m->journal[j].bank == e.bank) ok = false;
    }
    if (!ok) {
        const int n = m->journal_n + (m->journal_overflow ? 1 : 0);
        m->stats_counts.lost_calls += (uint64_t)n;
        const bool over = m->journal_overflow;
        journal_clear(m);
        return fail(APE_ERR_HIP, "%d pending call(s) could not be re-issued (%s) behind an aborted launch: %s", n,
                    over ? "more than 64 calls since the last check, or a call the journal cannot hold"
                         : "a stream bank has moved on since the aborted step", what.c_str());
    }
    m->replaying = true;
    int bad = APE_OK;
    for (int i = 0; i < m->journal_n && bad == APE_OK; ++i) {
        bad = replay_entry(m, m->journal[i]);
        if (bad == APE_OK) m->stats_counts.reissued_calls += 1;
    }
    m->replaying = false;
    const int n = m->journal_n;
    journal_clear(m);
    if (bad != APE_OK) { m->stats_counts.lost_calls += (uint64_t)n; return bad; }
    HIP_TRY(hipDeviceSynchronize());
    return APE_OK;
}

static int check_and_reset(ape_model_t* m) {
    if (!m) return fail(APE_ERR_INVALID_ARG, "check: NULL model");
    if (m->ffp_ok) {                                // the MLP pipeline's status word (bounded spins between its two stages)
        HIP_TRY(hipSetDevice(m->dims.device));
        HIP_TRY(hipDeviceSynchronize());
        unsigned st = 0;
        HIP_TRY(hipMemcpy(&st, m->ffp_ctl + 8 * 16, sizeof(st), hipMemcpyDeviceToHost));
        if (st != 0) {
            HIP_TRY(hipMemset(m->ffp_ctl, 0, m->ffp_ctl_words * sizeof(unsigned)));
            HIP_TRY(hipDeviceSynchronize());
            return fail(APE_ERR_HIP, "mlp pipeline launch aborted (status %u); outputs since the last successful check are invalid; the model "
                        "is usable again", st);
        }
    }
    if (!m->cluster_ok) return APE_OK;
    HIP_TRY(hipSetDevice(m->dims.device));
    // every stream of the device, non-blocking ones included: a launch that is still spinning must have ended (its
    // status word written) before the word is read, and nothing may be in flight when the flags are reset below
    HIP_TRY(hipDeviceSynchronize());
    unsigned st = 0;
    HIP_TRY(hipMemcpy(&st, m->xflags + m->xflag_bytes / sizeof(unsigned), sizeof(st), hipMemcpyDeviceToHost));
    if (st != 0) {
        // an aborted launch skipped its self-cleaning: reset flags, counters and the status word from the host
        HIP_TRY(hipMemset(m->xflags, 0, m->xflag_bytes + 16));
        HIP_TRY(hipMemset(m->xcc_slots, 0, APE_XCC_WORDS * sizeof(unsigned)));
        HIP_TRY(hipMemset(m->hxs, 0, 256 + m->hxs_bytes));     // the aborted launch's granules carry tags the next launch would await
        if (m->gxm) HIP_TRY(hipMemset(m->gxm, 0, 256 + 8 * m->gxm_cluster_bytes + 8 * 64 * 8));
        if (m->gx16) HIP_TRY(hipMemset(m->gx16, 0, 256 + m->gx16_bytes));
        HIP_TRY(hipDeviceSynchronize());
        return fail(APE_ERR_HIP, "cluster kernel launch aborted (status %u: %s); outputs of every launch on this model "
                    "since the last successful check are invalid; the model is usable again", st,
                    st == 1 ? "a workgroup gave up waiting for a peer -- not all workgroups of a cluster were resident"
                    : st == 5 ? "the Monte-Carlo latency kernel gave up waiting for the members of a cluster -- not all workgroups were resident"
                    : st >= 16 ? "the Monte-Carlo latency kernel gave up waiting for a peer's values (16 + phase)"
                            : "a launch found the state of an earlier aborted launch");
    }
    return APE_OK;
}

static int fk_impl(ape_model_t* m, const void* preds_dev, int32_t preds_dtype, int32_t N, int32_t denormalize, void* est_dev,
                   int32_t est_dtype, void* stream);

int ape_fk(ape_model_t* m, const void* preds_dev, int32_t preds_dtype, int32_t N, int32_t denormalize, void* est_dev,
           int32_t est_dtype, void* stream) {
    const int rc = fk_impl(m, preds_dev, preds_dtype, N, denormalize, est_dev, est_dtype, stream);
    if (rc == APE_OK) {      // (consumes what an aborted launch may have left unwritten: re-issued behind it)
        ApeJournalEntry e{};
        e.kind = ApeJournalEntry::FK; e.in0 = preds_dev; e.out0 = est_dev; e.B = N; e.i0 = preds_dtype; e.i1 = denormalize; e.i2 = est_dtype;
        e.stream = stream;
        journal_add(m, e);
    }
    return rc;
}

static int fk_impl(ape_model_t* m, const void* preds_dev, int32_t preds_dtype, int32_t N, int32_t denormalize, void* est_dev,
                   int32_t est_dtype, void* stream) {
    if (!m || !preds_dev || !est_dev) return fail(APE_ERR_INVALID_ARG, "fk: NULL argument");
    if (N < 1) return fail(APE_ERR_INVALID_ARG, "fk: N=%d must be >= 1", N);
    if ((preds_dtype != APE_F32 && preds_dtype != APE_F64) || (est_dtype != APE_F32 && est_dtype != APE_F64))
        return fail(APE_ERR_INVALID_ARG, "fk: unknown dtype selector");
    if (m->dims.target_layout == APE_LAYOUT_NONE) return fail(APE_ERR_INVALID_ARG, "fk: model has no target layout");
    if (denormalize && !m->has_stats) return fail(APE_ERR_NOT_READY, "fk: denormalize without norm stats");
    FkParams p{};
    p.preds = preds_dev;
    p.est = est_dev;
    if (denormalize) {
        p.yy_m = m->stats + 2 * m->dims.input_size;
        p.yy_s = p.yy_m + m->dims.output_size;
    }
    memcpy(p.body, m->body, sizeof(p.body));
    p.N = N; p.O = m->dims.output_size; p.layout = m->dims.target_layout; p.W = layout_est_width(p.layout);
    hipError_t e = ape_launch_fk(p, preds_dtype, est_dtype, (hipStream_t)stream);
    if (e != hipSuccess) return fail(APE_ERR_HIP, "fk kernel launch failed: %s", hipGetErrorString(e));
    return APE_OK;
}

int ape_msg_reduce(ape_model_t* m, const double* est_dev, int32_t N, double* msg_dev, void* stream) {
    if (!m || !est_dev || !msg_dev) return fail(APE_ERR_INVALID_ARG, "msg_reduce: NULL argument");
    if (N < 1) return fail(APE_ERR_INVALID_ARG, "msg_reduce: N=%d must be >= 1", N);
    if (m->dims.target_layout == APE_LAYOUT_NONE) return fail(APE_ERR_INVALID_ARG, "msg_reduce: model has no target layout");
    MsgParams p{};
    p.est = est_dev;
    p.msg = msg_dev;
    memcpy(p.body, m->body, sizeof(p.body));
    p.N = N; p.layout = m->dims.target_layout; p.W = layout_est_width(p.layout);
    hipError_t e = ape_launch_msg_reduce(p, (hipStream_t)stream);
    if (e != hipSuccess) return fail(APE_ERR_HIP, "msg kernel launch failed: %s", hipGetErrorString(e));
    ApeJournalEntry je{};
    je.kind = ApeJournalEntry::MSG; je.in0 = est_dev; je.out0 = msg_dev; je.B = N; je.stream = stream;
    journal_add(m, je);
    return APE_OK;
}

int ape_parse_rows(int32_t kind, const float* rows_dev, int32_t N, void* xx_dev, int32_t xx_dtype, void* stream) {
    if (!rows_dev || !xx_dev) return fail(APE_ERR_INVALID_ARG, "parse_rows: NULL argument");
    if (N < 1) return fail(APE_ERR_INVALID_ARG, "parse_rows: N=%d must be >= 1", N);
    if (xx_dtype != APE_F32 && xx_dtype != APE_F64) return fail(APE_ERR_INVALID_ARG, "parse_rows: unknown dtype selector");
    int width, I;
    const int big_endian = (kind & APE_PARSE_BIG_ENDIAN) ? 1 : 0;
    kind &= ~APE_PARSE_BIG_ENDIAN;
    if (!parse_kind_dims(kind, &width, &I)) return fail(APE_ERR_INVALID_ARG, "parse_rows: unknown kind %d", kind);
    hipError_t e = ape_launch_parse_rows(rows_dev, N, width, kind, xx_dev, xx_dtype, I, (size_t)I, 1, 0, big_endian,
                                         (hipStream_t)stream);
    if (e != hipSuccess) return fail(APE_ERR_HIP, "parse_rows launch failed: %s", hipGetErrorString(e));
    return APE_OK;
}

int ape_infer(ape_model_t* m, const float* x_dev, int32_t B, int32_t T, uint32_t flags, float* y_dev, void* est_dev,
              int32_t est_dtype, void* stream) {
    if (!m || !x_dev || !est_dev) return fail(APE_ERR_INVALID_ARG, "infer: NULL argument");
    if (flags & (APE_FLAG_ALL_STEPS | APE_FLAG_DROPOUT_MASKS | APE_FLAG_DROPOUT_PHILOX))
        return fail(APE_ERR_INVALID_ARG, "infer: only NORMALIZE_INPUT is accepted (use ape_lstm_forward + ape_fk)");
    float* y = y_dev;
    if (!y) {
        if (B > m->y_cap) {
            hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
            (void)hipStreamIsCapturing((hipStream_t)stream, &st);
            if (st != hipStreamCaptureStatusNone)
                return fail(APE_ERR_CAPACITY, "infer: B=%d exceeds reserved %d during stream capture", B, m->y_cap);
            if (int rc = ape_model_reserve(m, B)) return rc;
        }
        y = m->y_ws;
    }
    // de-normalise exactly when the inputs were normalised (estimator.py:103-109: one switch)
    const bool denorm = (flags & APE_FLAG_NORMALIZE_INPUT) != 0;
    if (est_dtype != APE_F32 && est_dtype != APE_F64) return fail(APE_ERR_INVALID_ARG, "fk: unknown dtype selector");
    if (m->dims.target_layout == APE_LAYOUT_NONE) return fail(APE_ERR_INVALID_ARG, "fk: model has no target layout");
    if (denorm && !m->has_stats) return fail(APE_ERR_NOT_READY, "fk: denormalize without norm stats");
    FkTail tail{est_dev, est_dtype, denorm, false};
    if (int rc = lstm_forward_impl(m, x_dev, B, T, flags, nullptr, 0.0f, 0, y, stream, 0, nullptr, nullptr, &tail)) return rc;
    if (!tail.done)                                        // (else the latency kernel finished the rows itself)
        if (int rc = fk_impl(m, y, APE_F32, B, denorm ? 1 : 0, est_dev, est_dtype, stream)) return rc;
    ApeJournalEntry e{};
    e.kind = ApeJournalEntry::INFER; e.in0 = x_dev; e.out0 = y_dev; e.out1 = est_dev; e.B = B; e.T = T; e.flags = flags; e.i2 = est_dtype;
    e.stream = stream;
    journal_add(m, e);
    return APE_OK;
}


// ---- stream bank ------------------------------------------------------------------------------------------

// (re)allocates the three rings for the bank's current S, T, smooth, n_mc
static hipError_t bank_alloc(ape_streams* b) {
    const size_t I = b->model->dims.input_size, O = b->model->dims.output_size, R = (size_t)b->S * b->n_mc;
    if (b->xring) (void)hipFree(b->xring);
    if (b->yring) (void)hipFree(b->yring);
    if (b->y_new) (void)hipFree(b->y_new);
    b->xring = b->yring = b->y_new = nullptr;
    hipError_t e = hipMalloc((void**)&b->xring, R * b->T * I * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&b->yring, R * b->smooth * O * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&b->y_new, R * O * sizeof(float));
    // a few streams with tall smoothing stacks: the post-filter deals a stream's stack over several workgroups (one CU each)
    if (b->post_part) (void)hipFree(b->post_part);
    b->post_part = nullptr; b->post_cnt = nullptr;
    const int chunks = ape_stream_post_chunks(b->smooth * b->n_mc);
    if (e == hipSuccess && chunks > 1 && (long long)b->S * chunks <= b->model->n_cus) {
        const size_t part_bytes = (size_t)b->S * chunks * 21 * sizeof(double);
        e = hipMalloc((void**)&b->post_part, part_bytes + (size_t)b->S * sizeof(unsigned));
        if (e == hipSuccess) e = hipMemset(b->post_part, 0, part_bytes + (size_t)b->S * sizeof(unsigned));
        if (e == hipSuccess) b->post_cnt = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(b->post_part) + part_bytes);
    }
    return e;
}

int ape_streams_create(ape_model_t* m, int32_t n_streams, int32_t seq_len, int32_t smooth, ape_streams_t** out) {
    if (!out) return fail(APE_ERR_INVALID_ARG, "streams_create: out is NULL");
    *out = nullptr;
    if (!m) return fail(APE_ERR_INVALID_ARG, "streams_create: NULL model");
    if (m->dims.model_kind != APE_MODEL_LSTM) return fail(APE_ERR_UNSUPPORTED, "streams_create: LSTM models only");
    if (m->dims.target_layout == APE_LAYOUT_NONE) return fail(APE_ERR_INVALID_ARG, "streams_create: model has no target layout");
    if (n_streams < 1 || seq_len < 1) return fail(APE_ERR_INVALID_ARG, "streams_create: n_streams=%d seq_len=%d", n_streams, seq_len);
    if (smooth < 1 || smooth > 64) return fail(APE_ERR_UNSUPPORTED, "streams_create: smooth %d outside 1..64", smooth);
    ape_streams* b = new (std::nothrow) ape_streams();
    if (!b) return fail(APE_ERR_HIP, "out of host memory");
    b->model = m; b->S = n_streams; b->T = seq_len; b->smooth = smooth;
    hipError_t e = hipSetDevice(m->dims.device);
    if (e == hipSuccess) e = bank_alloc(b);
    if (e != hipSuccess) {
        ape_streams_destroy(b);
        return fail(APE_ERR_HIP, "streams_create: allocation failed: %s", hipGetErrorString(e));
    }
    *out = b;
    return APE_OK;
}

// Sample rows per launch of a Monte-Carlo bank's weight-stationary route (pure arithmetic: ape_debug_bank_chunks, tests/test_plan_cpu.py).
// The pre-laid input is T x 1 KiB per sample row of a 2 x 256 model (lstm_upper32.hip), T x 512 B of the 3 x 128 model
// (lstm_upper128.hip); it and -- 2 x 256 models -- launch A's sequence and input tiles sit behind ONE 32-bit buffer descriptor each, so
// a bank that outgrows 2 GiB of any of them is chunked (launch B) or keeps the batch-tile route (launch A: 65 536 streams of 64 steps).
// Sample rows from which a Monte-Carlo bank of a 2 x 256 model takes the weight-stationary route (layer 0 once per stream, the layer above
// over the sample rows, both on lstm_upper32.hip): above 512 -- where the fused first-generation dropout kernel needs a second launch.
// (Until round 5: 2048, two tiles per cluster -- a cluster with ONE tile paid its exchange in the open; with the SOLO form it does not.
//  Measured, pocket, T = 6, frame of all streams: 21 x 25 rows 159 -> 117 us, 41 x 25 221 -> 160, 80 x 25 290 -> 168, 34 x 60 318 -> 167;
//  up to 512 rows the one fused launch stays ahead: 20 x 25 97.5 against 116.)
// ---- which route a Monte-Carlo bank takes, as pure arithmetic on (model shape, CU count, bank size): used by ape_streams_set_mc /
// streams_step_impl and, for the CPU tests of the thresholds, by ape_debug_bank_route ---------------------------------------------------
enum { BANK_FUSED = 0, BANK_SHARED_TILE16 = 1, BANK_UPPER32 = 2, BANK_UPPER128 = 3 };              // route of the layers above layer 0
enum { BANK_A_NONE = 0, BANK_A_TILE16 = 1, BANK_A_SEQ32 = 2, BANK_A_ONE_LAYER = 3 };                // kernel of launch A (layer 0 once per stream)
static bool bank_shares_layer0(long long sample_rows, int n_cus, bool can_up32, bool can_up128);
static int bank_launch_a_form(int route, int H, int KX, int n_cus, int S, bool cluster_ok);
#ifndef APE_BANK_SHARE_MIN_ROWS_128
// the 3 x 128 model's route (lstm_upper128.hip; no one-tile form, every exchange of a one-tile cluster is exposed): from where the fused
// first-generation dropout kernel needs a THIRD launch.  Measured, T = 6, frame of all streams, fused launches / this route: 11 x 50 rows
// 125.4 / 137.9 us -- 21 x 50 213.8 / 141.6, 30 x 50 213.8 / 146.1, 40 x 50 215.8 / 150.2, 64 x 25 179.9 / 147.2 (2048 until round 5)
#define APE_BANK_SHARE_MIN_ROWS_128 1025
#endif
#ifndef APE_BANK_A_ONE_LAYER_MAX_STREAMS
// 2 x 256 banks: launch A on the first-generation kernel's one-layer form up to this many streams (three any-placement clusters of 32;
// from four clusters on that kernel forms XCD classes and the rendezvous eats the gain).  Measured, pocket, T = 6, frame of all streams,
// one-layer form against the SEQ form of lstm_upper32.hip: 21 x 25 108.1 / 118.7 us, 41 x 25 155.3 / 165.7, 64 x 25 157.2 / 165.2,
// 80 x 25 173.0 / 178.7 -- 100 x 25 220.3 / 215.8, 200 x 25 341.5 / 338.3, 512 x 25 767.4 / 748.4.
#define APE_BANK_A_ONE_LAYER_MAX_STREAMS 96
#endif
#ifndef APE_BANK_SHARE_MIN_ROWS
#define APE_BANK_SHARE_MIN_ROWS 513
#endif
static bool bank_chunk_plan(bool up128, int S, int T, int n_mc, long long* chunk_rows) {
    const long long total = (long long)S * n_mc;
    const long long max_chunk = ((2047ll << 20) / ((long long)T * (up128 ? 512 : 1024))) / 1024 * 1024;
    const bool l0_fits = up128 || (ape_lower32_hseq_bytes(S, T) < (2047ull << 20) && ape_lower32_xfrag_bytes(S, T) < (2047ull << 20));
    if (max_chunk < 1024 || total >= (1ll << 31) || !l0_fits) return false;       // (the input builder indexes sample rows with 32 bits)
    const long long n_chunks = (total + max_chunk - 1) / max_chunk;
    long long chunk = ((total + n_chunks - 1) / n_chunks + 1023) / 1024 * 1024;
    if (chunk > total) chunk = (total + 31) / 32 * 32;
    *chunk_rows = chunk;
    return true;
}

static bool bank_shares_layer0(long long sample_rows, int n_cus, bool can_up32, bool can_up128) {
    return sample_rows >= 2LL * tile16_wave_rows(n_cus) ||
           (can_up32 && sample_rows >= (can_up128 ? APE_BANK_SHARE_MIN_ROWS_128 : APE_BANK_SHARE_MIN_ROWS));
}
// launch A of a bank on `route` (a cooperative step on a healthy model; a replay or a forced kernel takes the batch-tile launch)
static int bank_launch_a_form(int route, int H, int KX, int n_cus, int S, bool cluster_ok) {
    if (route == BANK_FUSED) return BANK_A_NONE;
    const bool one_layer_fits = cluster_ok && ape_cluster_layer0_supported(H, KX) && (S + 31) / 32 <= cluster_capacity(n_cus, H);
    if (route == BANK_UPPER32) return (one_layer_fits && S <= APE_BANK_A_ONE_LAYER_MAX_STREAMS) ? BANK_A_ONE_LAYER : BANK_A_SEQ32;
    if (route == BANK_UPPER128) return one_layer_fits ? BANK_A_ONE_LAYER : BANK_A_TILE16;
    return BANK_A_TILE16;
}

int ape_streams_set_mc(ape_streams_t* b, int32_t n_mc, float dropout_p, uint64_t seed) {
    if (!b) return fail(APE_ERR_INVALID_ARG, "streams_set_mc: NULL bank");
    if (n_mc < 1 || (long long)n_mc * b->smooth > 4096)
        return fail(APE_ERR_INVALID_ARG, "streams_set_mc: n_mc=%d with smooth=%d (1 <= smooth*n_mc <= 4096)", n_mc, b->smooth);
    if (!(dropout_p >= 0.0f && dropout_p < 1.0f)) return fail(APE_ERR_INVALID_ARG, "streams_set_mc: dropout_p %g outside [0,1)", (double)dropout_p);
    HIP_TRY(hipSetDevice(b->model->dims.device));
    HIP_TRY(hipDeviceSynchronize());             // the rings may still be read by an earlier step
    b->n_mc = n_mc; b->mc = true; b->dropout_p = dropout_p; b->seed = seed; b->mc_calls = 0;
    b->frames = 0; b->steps = 0;
    hipError_t e = bank_alloc(b);
    if (e != hipSuccess) {
        // the old rings are gone and the new ones are incomplete: the bank stays unusable (every push / step refuses)
        // until a later ape_streams_set_mc succeeds
        if (b->xring) (void)hipFree(b->xring);
        if (b->yring) (void)hipFree(b->yring);
        if (b->y_new) (void)hipFree(b->y_new);
        b->xring = b->yring = b->y_new = nullptr;
        return fail(APE_ERR_HIP, "streams_set_mc: allocation failed: %s", hipGetErrorString(e));
    }
    // Layer 0 once per stream (nn.LSTM's dropout sits BETWEEN the layers, so h_0(t) is the same for all samples of a
    // stream): worth its extra launch from two batch-tile waves of sample rows on.  The [S,T,H] sequence lives in the
    // model's all-steps workspace, sized here so that the step itself never allocates.
    ape_model* m = b->model;
    // (on the weight-stationary route -- lstm_upper32.hip, both launches -- the sharing pays from APE_BANK_SHARE_MIN_ROWS sample rows on;
    //  the 3 x 128 model's route, lstm_upper128.hip, from APE_BANK_SHARE_MIN_ROWS_128)
    const long long sample_rows = (long long)b->S * n_mc;
    const bool can_up128 = m->up128_ok && m->c32_on;
    const bool can_up32 = (m->up32_ok && m->c32_on && f16v2_capacity(m->n_cus) >= 8) || can_up128;
    b->shared_l0 = m->upper_ok && m->kernel_choice == APE_KERNEL_AUTO && m->precision == APE_PRECISION_F32 &&
                   dropout_p > 0.0f && n_mc >= 2 &&
                   bank_shares_layer0(sample_rows, m->n_cus, can_up32, can_up128);
    if (b->xfrag) { (void)hipFree(b->xfrag); b->xfrag = nullptr; }
    if (b->ypart) { (void)hipFree(b->ypart); b->ypart = nullptr; }
    if (b->xfrag0) { (void)hipFree(b->xfrag0); b->xfrag0 = nullptr; }
    if (b->hfrag) { (void)hipFree(b->hfrag); b->hfrag = nullptr; }
    b->up32 = false;
    b->up128 = false;
    if (b->maskbits) { (void)hipFree(b->maskbits); b->maskbits = nullptr; }
    if (b->shared_l0) {
        const size_t rows = (size_t)b->S * b->T;
        if (rows > m->hseq_cap) {
            if (m->hseq_ws) { HIP_TRY(hipFree(m->hseq_ws)); m->hseq_ws = nullptr; m->hseq_cap = 0; }
            HIP_TRY(hipMalloc((void**)&m->hseq_ws, rows * m->dims.hidden_size * sizeof(float)));
            m->hseq_cap = rows;
        }
        if (can_up32) {
            // chunks of equal size whose expanded input (T KiB per sample row) stays under 2 GiB -- inside one 32-bit buffer
            // descriptor with offsets to spare; whole 1024-row waves of clusters where that costs nothing.  (Measured at
            // 8192 x 25, T = 6: one 1.26 GB chunk 9.30 ms per frame, five 256 MB chunks -- the Infinity Cache's size -- 9.38 ms: a
            // launch's prologue and tail cost more than the cache residency of the tiles buys; 288 GB of HBM make the footprint
            // a non-issue.)
            long long chunk = 0;
            if (bank_chunk_plan(can_up128, b->S, b->T, n_mc, &chunk)) {
                b->chunk_rows = (int)chunk;
                if (can_up128) {
                    // the 3 x 128 model: launch A stays on the batch-tile kernel ([S,T,H] in the model's sequence workspace), launch B =
                    // layers 1 and 2 on lstm_upper128.hip
                    HIP_TRY(hipMalloc((void**)&b->xfrag, ape_upper128_xfrag_bytes(b->chunk_rows, b->T)));
                    HIP_TRY(hipMalloc((void**)&b->ypart, ape_upper128_ypart_bytes(b->chunk_rows)));
                    HIP_TRY(hipMalloc((void**)&b->maskbits, ape_upper128_maskbits_bytes(b->chunk_rows, b->T)));
                    b->up128 = true;
                } else {
                HIP_TRY(hipMalloc((void**)&b->xfrag, ape_upper32_xfrag_bytes(b->chunk_rows, b->T)));
                HIP_TRY(hipMalloc((void**)&b->ypart, ape_upper32_ypart_bytes(b->chunk_rows)));
                // layer 0 on the same cluster structure (its SEQ form): input tiles and the [tile][step] sequence, fragment order
                HIP_TRY(hipMalloc((void**)&b->xfrag0, ape_lower32_xfrag_bytes(b->S, b->T)));
                HIP_TRY(hipMalloc((void**)&b->hfrag, ape_lower32_hseq_bytes(b->S, b->T)));
                b->up32 = true;
                }
            }
        }
    }
    return APE_OK;
}

int ape_streams_destroy(ape_streams_t* b) {
    if (!b) return APE_OK;
    if (b->xring) (void)hipFree(b->xring);
    if (b->yring) (void)hipFree(b->yring);
    if (b->y_new) (void)hipFree(b->y_new);
    if (b->post_part) (void)hipFree(b->post_part);
    if (b->xfrag) (void)hipFree(b->xfrag);
    if (b->ypart) (void)hipFree(b->ypart);
    if (b->xfrag0) (void)hipFree(b->xfrag0);
    if (b->hfrag) (void)hipFree(b->hfrag);
    if (b->maskbits) (void)hipFree(b->maskbits);
    if (b->h_rows) (void)hipHostFree(b->h_rows);
    if (b->h_out) (void)hipHostFree(b->h_out);
    if (b->h_status) (void)hipHostFree(b->h_status);
    if (b->h_done) (void)hipHostFree(b->h_done);
    for (auto ev : b->prof_ev) if (ev) (void)hipEventDestroy(ev);
    if (ape_model* m = b->model) {      // pending steps of this bank can no longer be re-issued
        int k = 0;
        for (int i = 0; i < m->journal_n; ++i) {
            if (m->journal[i].kind == ApeJournalEntry::STEP && m->journal[i].bank == b) { m->journal_overflow = true; continue; }
            m->journal[k++] = m->journal[i];
        }
        m->journal_n = k;
    }
    delete b;
    return APE_OK;
}

int ape_streams_reset(ape_streams_t* b) {
    if (!b) return fail(APE_ERR_INVALID_ARG, "streams_reset: NULL bank");
    b->frames = 0; b->steps = 0;
    return APE_OK;
}

// where the next row goes: one slot of each of the stream's n_mc windows (stride T*I apart), or -- first row after
// a reset -- all n_mc*T slots, which are contiguous
static void next_slot(const ape_streams* b, size_t I, float** out, int* rep, size_t* rep_stride) {
    const bool cold = b->frames == 0;
    *out = b->xring + (cold ? 0 : (size_t)(b->frames % b->T) * I);
    const int copies = b->shared_l0 ? 1 : b->n_mc;      // layer 0 shared: only a stream's first window copy is ever read
    *rep = cold ? b->T * copies : copies;
    *rep_stride = cold ? I : (size_t)b->T * I;
}

int ape_streams_push_rows(ape_streams_t* b, int32_t kind, const float* rows_dev, void* stream) {
    if (!b || !rows_dev) return fail(APE_ERR_INVALID_ARG, "streams_push_rows: NULL argument");
    if (!b->xring) return fail(APE_ERR_NOT_READY, "streams_push_rows: the bank lost its rings in a failed ape_streams_set_mc");
    int width, I;
    const int big_endian = (kind & APE_PARSE_BIG_ENDIAN) ? 1 : 0;
    kind &= ~APE_PARSE_BIG_ENDIAN;
    if (!parse_kind_dims(kind, &width, &I)) return fail(APE_ERR_INVALID_ARG, "streams_push_rows: unknown kind %d", kind);
    if (I != b->model->dims.input_size)
        return fail(APE_ERR_INVALID_ARG, "streams_push_rows: kind %d builds %d features, the model takes %d", kind, I,
                    b->model->dims.input_size);
    float* out; int rep; size_t rep_stride;
    next_slot(b, (size_t)I, &out, &rep, &rep_stride);
    hipError_t e = ape_launch_parse_rows(rows_dev, b->S, width, kind, out, APE_F32, I, (size_t)b->n_mc * b->T * I, rep,
                                         rep_stride, big_endian, (hipStream_t)stream);
    if (e != hipSuccess) return fail(APE_ERR_HIP, "streams_push_rows launch failed: %s", hipGetErrorString(e));
    ++b->frames;
    return APE_OK;
}

int ape_streams_push_features(ape_streams_t* b, const float* xx_dev, void* stream) {
    if (!b || !xx_dev) return fail(APE_ERR_INVALID_ARG, "streams_push_features: NULL argument");
    if (!b->xring) return fail(APE_ERR_NOT_READY, "streams_push_features: the bank lost its rings in a failed ape_streams_set_mc");
    const int I = b->model->dims.input_size;
    float* out; int rep; size_t rep_stride;
    next_slot(b, (size_t)I, &out, &rep, &rep_stride);
    hipError_t e = ape_launch_ring_write(xx_dev, b->S, I, out, (size_t)b->n_mc * b->T * I, rep, rep_stride, (hipStream_t)stream);
    if (e != hipSuccess) return fail(APE_ERR_HIP, "streams_push_features launch failed: %s", hipGetErrorString(e));
    ++b->frames;
    return APE_OK;
}

struct McFrameParts;
static int streams_step_impl(ape_streams_t* b, uint32_t flags, void* msg_dev, void* tail_dev, int32_t out_dtype, void* stream,
                             unsigned* status_out, const McFrameParts* fuse = nullptr);

int ape_streams_step(ape_streams_t* b, uint32_t flags, void* msg_dev, void* tail_dev, int32_t out_dtype, void* stream) {
    return streams_step_impl(b, flags, msg_dev, tail_dev, out_dtype, stream, nullptr);
}

// status_out (host frames): pinned word that receives the model's sticky status word behind the step's kernels
// a few streams in Monte-Carlo mode (one estimator's frame: S = 1) step on the Monte-Carlo latency kernel
static bool bank_on_mc_small(const ape_streams* b) {
    const ape_model* m = b->model;
    return !b->shared_l0 && b->mc && b->dropout_p > 0.0f && m->dims.num_layers > 1 && b->S <= 8 && m->cluster_ok && b->T <= 64 &&
           mc_small_fits(m, b->S, b->n_mc);
}

static int streams_step_impl(ape_streams_t* b, uint32_t flags, void* msg_dev, void* tail_dev, int32_t out_dtype, void* stream,
                             unsigned* status_out, const McFrameParts* fuse) {
    if (!b || !msg_dev) return fail(APE_ERR_INVALID_ARG, "streams_step: NULL argument");
    if (!b->xring || !b->yring || !b->y_new)
        return fail(APE_ERR_NOT_READY, "streams_step: the bank lost its rings in a failed ape_streams_set_mc");
    if (b->frames == 0) return fail(APE_ERR_NOT_READY, "streams_step: no row pushed since the last reset");
    // the exchange-form selectors of include/ape_hip.h travel to the frame's LSTM launches unchanged (same bits whichever is set)
    const uint32_t diag_wt = flags & (APE_FLAG_ANY_PLACEMENT | APE_FLAG_IN_XCD_PLAIN | APE_FLAG_NO_XCD_CLASSES);
    flags &= ~diag_wt;
    if (flags & ~(uint32_t)(APE_FLAG_NORMALIZE_INPUT | APE_FLAG_PACKED_MSG))
        return fail(APE_ERR_INVALID_ARG, "streams_step: NORMALIZE_INPUT, PACKED_MSG and the exchange-form selectors ANY_PLACEMENT, "
                    "IN_XCD_PLAIN, NO_XCD_CLASSES are accepted");
    if (out_dtype != APE_F32 && out_dtype != APE_F64) return fail(APE_ERR_INVALID_ARG, "streams_step: unknown dtype selector");
    const bool packed = (flags & APE_FLAG_PACKED_MSG) != 0;
    if (packed && tail_dev)
        return fail(APE_ERR_INVALID_ARG, "streams_step: PACKED_MSG rows carry the tail themselves (tail_dev must be NULL)");
    flags &= ~(uint32_t)APE_FLAG_PACKED_MSG;
    ape_model* m = b->model;
    const bool norm = (flags & APE_FLAG_NORMALIZE_INPUT) != 0;
    if (norm && !m->has_stats) return fail(APE_ERR_NOT_READY, "streams_step: NORMALIZE_INPUT without norm stats");
    // oldest row of every window: the slot after the newest one
    const int x_ring = (int)(b->frames % b->T);
    const bool drop = b->mc && b->dropout_p > 0.0f && m->dims.num_layers > 1;
    ApeJournalEntry je{};
    je.kind = ApeJournalEntry::STEP; je.out0 = msg_dev; je.out1 = tail_dev; je.flags = flags | (packed ? APE_FLAG_PACKED_MSG : 0u);
    je.i2 = out_dtype; je.stream = stream; je.bank = b; je.bank_frames = b->frames; je.bank_steps = b->steps; je.bank_mc_calls = b->mc_calls;
    // measurement aid (ape_streams_profile): a pair of events around every launch of the step's dominant kernel
    auto prof_pair = [&](hipEvent_t* a, hipEvent_t* z) {
        *a = *z = nullptr;
        if (!b->prof_on || b->prof_n >= (int)b->prof_ev.size() / 2) return;
        *a = b->prof_ev[2 * b->prof_n]; *z = b->prof_ev[2 * b->prof_n + 1];
        b->prof_n += 1;
    };
    auto post_params = [&]() {
        StreamPostParams q{};
        q.y_new = b->y_new; q.yring = b->yring; q.msg = msg_dev; q.tail = tail_dev;
        q.yy_m = norm ? m->stats + 2 * m->dims.input_size : nullptr;      // one switch (estimator.py:103-109)
        q.yy_s = norm ? m->stats + 2 * m->dims.input_size + m->dims.output_size : nullptr;
        memcpy(q.body, m->body, sizeof(q.body));
        q.S = b->S; q.O = m->dims.output_size; q.W = layout_est_width(m->dims.target_layout); q.layout = m->dims.target_layout;
        q.smooth = b->smooth; q.n_mc = b->n_mc;
        q.pos = (int)(b->steps % b->smooth);
        q.cold = b->steps == 0 ? 1 : 0;
        q.msg_dtype = out_dtype; q.packed = packed ? 1 : 0;
        if (status_out != nullptr && m->cluster_ok) { q.status_in = m->xflags + m->xflag_bytes / sizeof(unsigned); q.status_out = status_out; }
        if (status_out != nullptr && b->h_done != nullptr) { q.done_out = b->h_done; q.done_val = b->h_done_val; }
        q.part = b->post_part; q.part_cnt = b->post_cnt;
        return q;
    };
    if (b->shared_l0) {
        if (!m->has_weights) return fail(APE_ERR_NOT_READY, "streams_step: weights not loaded");
        if ((size_t)b->S * b->T > m->hseq_cap) return fail(APE_ERR_CAPACITY, "streams_step: the layer-0 sequence workspace is gone");
        const int H = m->dims.hidden_size, I = m->dims.input_size, O = m->dims.output_size;
        // (the cooperative routes were planned by ape_streams_set_mc; a later set_kernel(TILE16 / AUTO_GEN1 / CLUSTER) or set_precision --
        //  the documented ways to keep persistent clusters off a shared GPU -- sends the step to the batch-tile route like a replay)
        const bool coop = !m->replaying && m->kernel_choice == APE_KERNEL_AUTO && m->c32_on && m->precision == APE_PRECISION_F32;
        const bool cluster_route = b->up32 && coop;
        hipError_t e = hipSuccess;
        // launch A on the first-generation cluster kernel's one-layer form (lstm_cluster.hip <H, 1, KX, 2>): 32 streams per cluster of H / 16
        // members, every step's output -> [S,T,H] in the model's sequence workspace, as the batch-tile launch writes it
        auto launch_a_one_layer = [&]() -> hipError_t {
            ClusterParams c{};
            c.x = b->xring; c.x_row_stride = (size_t)b->n_mc * b->T * I;
            c.y = nullptr; c.hseq = m->hseq_ws;
            c.wcl[0] = m->wcl[0]; c.bias[0] = m->bias[0];
            c.w_out = m->w_out; c.b_out = m->b_out;
            c.xx_m = m->stats; c.xx_s = m->stats + I; c.xx_r = m->stats + 2 * I + 2 * O;
            c.hx = m->hx; c.hx_bytes = m->hx_bytes;
            c.xflags = m->xflags; c.status = m->xflags + m->xflag_bytes / sizeof(unsigned);
            c.ticket = c.status - 4; c.done = c.ticket + 1;
            c.B = b->S; c.T = b->T; c.I = I; c.O = O; c.x_ring = x_ring;
            c.flags = (flags & APE_FLAG_NORMALIZE_INPUT) | diag_wt;
            c.dbg_wg = m->dbg_wg; c.xcc_slots = m->xcc_slots;
            int clusters = (b->S + 31) / 32;
            if (m->gen1_classes && !(diag_wt & APE_FLAG_NO_XCD_CLASSES)) {
                const int c8 = (clusters + 7) / 8 * 8;
                if (clusters >= 4 && c8 <= cluster_capacity(m->n_cus, H)) { clusters = c8; c.flags |= APE_FLAG_XCD_CLASSES; }
            }
            return ape_launch_lstm_cluster(H, 1, m->KX, 2, false, clusters, c, (hipStream_t)stream);
        };
        const int a_form = !coop ? BANK_A_TILE16
                                 : bank_launch_a_form(b->up32 ? BANK_UPPER32 : b->up128 ? BANK_UPPER128 : BANK_SHARED_TILE16, H, m->KX, m->n_cus, b->S, m->cluster_ok);
        // (2 x 256 models, small banks: the one-layer form's 16-member clusters have half the matrix work per CU and step of the 8-member
        //  SEQ form below and the same exposed exchange; APE_BANK_A_ONE_LAYER_MAX_STREAMS above)
        const bool a_on_gen1 = cluster_route && a_form == BANK_A_ONE_LAYER;
        if (a_on_gen1) {
            e = launch_a_one_layer();
            if (e != hipSuccess) return fail(APE_ERR_HIP, "streams_step: layer-0 cluster launch failed: %s", hipGetErrorString(e));
        } else if (cluster_route) {
            // launch A on the weight-stationary structure (lstm_upper32.hip, SEQ form): S streams in tiles of 32 on the clusters,
            // every step's output in the fragment order launch B's input builder reads
            XFragParams xq{};
            xq.x = b->xring; xq.xfrag = b->xfrag0; xq.x_row_stride = (size_t)b->n_mc * b->T * I;
            xq.xx_m = norm ? m->stats : nullptr; xq.xx_s = norm ? m->stats + I : nullptr;
            xq.S = b->S; xq.T = b->T; xq.I = I; xq.x_ring = x_ring;
            UpperParams u{};
            u.xfrag = b->xfrag0; u.xfrag_bytes = ape_lower32_xfrag_bytes(b->S, b->T);
            u.w = m->wcl32[0]; u.bias = m->bias[0]; u.w_out = m->w_out;
            u.hx = m->hx; u.hx_bytes = m->hx_bytes;
            u.xflags = m->xflags; u.status = m->xflags + m->xflag_bytes / sizeof(unsigned); u.done = u.status - 3;
            u.xcc_slots = m->xcc_slots; u.dbg_wg = m->dbg_wg;
            u.hseq = b->hfrag; u.hseq_bytes = ape_lower32_hseq_bytes(b->S, b->T);
            u.T = b->T; u.O = O; u.n_tiles = (b->S + 31) / 32;
            u.flags = diag_wt;
            e = ape_launch_lstm_lower32(u, xq, f16v2_capacity(m->n_cus), (hipStream_t)stream);
            if (e != hipSuccess) return fail(APE_ERR_HIP, "streams_step: layer-0 cluster launch failed: %s", hipGetErrorString(e));
        } else if (b->up128 && a_form == BANK_A_ONE_LAYER) {
            // launch A of the 3 x 128 bank on the one-layer form (round 5: 1024 streams are 64 tiles of the batch-tile kernel -- a quarter
            // of the chip for 54 us; here every CU holds 16 units of a cluster: 30.8 us).  Every bank size that fits the chip's clusters:
            // 21 x 50 rows 141.6 -> 121.0 us per frame, 100 x 25 188.1 -> 164.5, 127 x 25 190.2 -> 166.4 against the batch-tile launch.
            e = launch_a_one_layer();
            if (e != hipSuccess) return fail(APE_ERR_HIP, "streams_step: layer-0 cluster launch failed: %s", hipGetErrorString(e));
        } else {
        // launch A: layer 0 alone over the S windows (first copy of every stream's ring), all steps -> [S,T,H]
        LstmParams a{};
        a.x = b->xring; a.x_row_stride = (size_t)b->n_mc * b->T * I;
        a.y = nullptr; a.hseq = m->hseq_ws;
        a.wpack[0] = m->wpack[0]; a.bias[0] = m->bias[0];
        a.w_out = m->w_out; a.b_out = m->b_out;
        a.xx_m = m->stats; a.xx_s = m->stats + I;
        a.B = b->S; a.T = b->T; a.I = I; a.O = O; a.KX = m->KX; a.x_ring = x_ring;
        a.flags = flags & APE_FLAG_NORMALIZE_INPUT;
        e = ape_launch_lstm_tile16(H, 1, a, (hipStream_t)stream);
        if (e != hipSuccess) return fail(APE_ERR_HIP, "streams_step: layer-0 launch failed: %s", hipGetErrorString(e));
        }
        // launch B: the layers above as an LSTM of their own over the S x n_mc sample rows; row r reads stream
        // r / n_mc's sequence under its own Philox mask (the counters of a fused launch over the same rows)
        if (cluster_route) {
            // weight-stationary form (lstm_upper32.hip): per chunk of sample rows the masked input in fragment order, the
            // persistent cluster kernel, the head reduce
            const long long total = (long long)b->S * b->n_mc;
            for (long long r0 = 0; r0 < total; r0 += b->chunk_rows) {
                const int rows = (int)((total - r0 < b->chunk_rows) ? total - r0 : b->chunk_rows);
                ExpandParams xq{};
                xq.hseq = a_on_gen1 ? m->hseq_ws : b->hfrag; xq.hseq_frag = a_on_gen1 ? 0 : 1;
                xq.xfrag = b->xfrag; xq.row_base = r0; xq.rows = rows; xq.T = b->T; xq.n_mc = b->n_mc;
                xq.layer = 0; xq.dropout_p = b->dropout_p; xq.seed = b->seed + b->mc_calls;
                xq.masks = b->inj_masks; xq.masks_rows = total;
                UpperParams u{};
                u.xfrag = b->xfrag; u.xfrag_bytes = ape_upper32_xfrag_bytes(rows, b->T); u.ypart = b->ypart;
                u.w = m->wcl32[1]; u.bias = m->bias[1]; u.w_out = m->w_out;
                u.hx = m->hx; u.hx_bytes = m->hx_bytes;
                u.xflags = m->xflags; u.status = m->xflags + m->xflag_bytes / sizeof(unsigned); u.done = u.status - 3;
                u.xcc_slots = m->xcc_slots; u.dbg_wg = m->dbg_wg;
                u.T = b->T; u.O = O; u.n_tiles = (rows + 31) / 32;
                u.flags = diag_wt;
#ifdef APE_ABLATE
                // timing experiments of the ablation library only (tests/tools/ablate_upper32.py; results are garbage)
                if (const char* ab = getenv("APE_UP32_ABLATE")) u.flags = (unsigned)strtoul(ab, nullptr, 0);
#endif
                hipEvent_t ev_a, ev_z;
                prof_pair(&ev_a, &ev_z);
                m->last_kernel = "ape_lstm_upper32";
                e = ape_launch_lstm_upper32(u, xq, m->b_out, b->y_new + (size_t)r0 * O, f16v2_capacity(m->n_cus), (hipStream_t)stream,
                                            ev_a, ev_z);
                if (e != hipSuccess) return fail(APE_ERR_HIP, "streams_step: upper-layer cluster launch failed: %s", hipGetErrorString(e));
            }
            ++b->mc_calls;
        } else if (b->up128 && coop) {
            // the 3 x 128 model: layers 1 and 2 on the four-member weight-stationary clusters (lstm_upper128.hip), per chunk of sample
            // rows: masked input + layer-1 keep bits, the persistent kernel, the head reduce
            const long long total = (long long)b->S * b->n_mc;
            for (long long r0 = 0; r0 < total; r0 += b->chunk_rows) {
                const int rows = (int)((total - r0 < b->chunk_rows) ? total - r0 : b->chunk_rows);
                ExpandParams xq{};
                xq.hseq = m->hseq_ws; xq.hseq_frag = 0; xq.xfrag = b->xfrag; xq.row_base = r0; xq.rows = rows; xq.T = b->T; xq.n_mc = b->n_mc;
                xq.layer = 0; xq.dropout_p = b->dropout_p; xq.seed = b->seed + b->mc_calls;
                xq.masks = b->inj_masks; xq.masks_rows = total;      // (test hooks: the caller's masks for both mask layers)
                Upper128Params u{};
                u.xfrag = b->xfrag; u.xfrag_bytes = ape_upper128_xfrag_bytes(rows, b->T); u.maskbits = b->maskbits; u.ypart = b->ypart;
                u.w[0] = m->wup128[1]; u.w[1] = m->wup128[2]; u.bias[0] = m->bias[1]; u.bias[1] = m->bias[2]; u.w_out = m->w_out;
                u.hx = m->hx; u.hx_bytes = m->hx_bytes;
                u.xflags = m->xflags; u.status = m->xflags + m->xflag_bytes / sizeof(unsigned); u.done = u.status - 3;
                u.xcc_slots = m->xcc_slots;
                u.T = b->T; u.O = O; u.n_tiles = (rows + 31) / 32;
                u.flags = diag_wt; u.dropout_p = b->dropout_p; u.dbg_wg = m->dbg_wg;
                hipEvent_t ev_a, ev_z;
                prof_pair(&ev_a, &ev_z);
                m->last_kernel = "ape_lstm_upper128";
                e = ape_launch_lstm_upper128(u, xq, m->b_out, b->y_new + (size_t)r0 * O, (m->n_cus / 4) / 8 * 8, (hipStream_t)stream, ev_a, ev_z);
                if (e != hipSuccess) return fail(APE_ERR_HIP, "streams_step: upper-layer cluster launch failed: %s", hipGetErrorString(e));
            }
            ++b->mc_calls;
        } else {
        if (b->inj_masks) return fail(APE_ERR_UNSUPPORTED, "streams_step: injected masks (test hook) on the batch-tile shared-layer-0 route");
        LstmParams q{};
        const int LU = m->dims.num_layers - 1;
        q.x = m->hseq_ws; q.y = b->y_new;
        for (int j = 0; j < LU; ++j) { q.wpack[j] = m->wpack[j + 1]; q.bias[j] = m->bias[j + 1]; }
        q.w_out = m->w_out; q.b_out = m->b_out;
        q.B = b->S * b->n_mc; q.T = b->T; q.I = H; q.O = O; q.KX = H; q.x_ring = 0;
        q.flags = APE_FLAG_DROPOUT_PHILOX; q.dropout_p = b->dropout_p; q.seed = b->seed + b->mc_calls;
        q.x_group = b->n_mc; q.layer_base = 1;
        hipEvent_t ev_a, ev_z;
        prof_pair(&ev_a, &ev_z);
        if (ev_a) (void)hipEventRecord(ev_a, (hipStream_t)stream);
        m->last_kernel = "ape_lstm_tile16";
        e = ape_launch_lstm_tile16(H, LU, q, (hipStream_t)stream);
        if (ev_z) (void)hipEventRecord(ev_z, (hipStream_t)stream);
        if (e != hipSuccess) return fail(APE_ERR_HIP, "streams_step: upper-layer launch failed: %s", hipGetErrorString(e));
        ++b->mc_calls;
        }
    } else if (bank_on_mc_small(b)) {
        // a few streams in Monte-Carlo mode (one estimator's frame: S = 1): the latency kernel reads the first copy of every
        // stream's window and deals the sample rows over the XCDs; from ape_streams_frame_host the feature builder rides in the same
        // launch as extra workgroups (`fuse`: the row has NOT been pushed by a launch of its own)
        if (!m->has_weights) return fail(APE_ERR_NOT_READY, "streams_step: weights not loaded");
        McFrameParts fr{};
        if (fuse) fr = *fuse;
        hipEvent_t ev_a, ev_z;
        prof_pair(&ev_a, &ev_z);
        if (ev_a) (void)hipEventRecord(ev_a, (hipStream_t)stream);
        if (int rc = mc_small_launch(m, b->xring, (size_t)b->n_mc * b->T * m->dims.input_size, b->S, b->n_mc, b->T,
                                     flags | diag_wt | (b->inj_masks ? APE_FLAG_DROPOUT_MASKS : APE_FLAG_DROPOUT_PHILOX), b->inj_masks,
                                     b->dropout_p, b->seed + b->mc_calls, b->y_new, stream, x_ring, &fr))
            return rc;
        if (ev_z) (void)hipEventRecord(ev_z, (hipStream_t)stream);
        ++b->mc_calls;
    } else {
        hipEvent_t ev_a, ev_z;
        prof_pair(&ev_a, &ev_z);
        if (ev_a) (void)hipEventRecord(ev_a, (hipStream_t)stream);
        if (int rc = lstm_forward_impl(m, b->xring, b->S * b->n_mc, b->T,
                                       flags | diag_wt | (!drop ? 0u : b->inj_masks ? APE_FLAG_DROPOUT_MASKS : APE_FLAG_DROPOUT_PHILOX),
                                       drop ? b->inj_masks : nullptr, drop ? b->dropout_p : 0.0f, b->seed + b->mc_calls, b->y_new, stream, x_ring))
            return rc;
        if (ev_z) (void)hipEventRecord(ev_z, (hipStream_t)stream);
        ++b->mc_calls;
    }
    StreamPostParams q = post_params();
    hipError_t e = ape_launch_stream_post(q, (hipStream_t)stream);
    if (e != hipSuccess) return fail(APE_ERR_HIP, "streams_step launch failed: %s", hipGetErrorString(e));
    ++b->steps;
    journal_add(m, je);
    return APE_OK;
}

// One iteration of Estimator.processing_loop (estimator.py:174-177) for every stream of the bank, HOST buffers in and out.
// The raw rows are copied into pinned, device-visible staging the feature builder reads directly; the post kernel writes the
// datagram rows and the model's status word straight into pinned host memory: the frame holds no copy command, just the
// step's kernels and one stream synchronisation.  An aborted cooperative launch is recovered here (ape_model_recover).
// The frame call polls words the device writes into pinned memory (h_done, h_status) and reads h_out without a stream synchronisation:
// the buffers are allocated COHERENT (fine-grained) and mapped explicitly -- with hipHostMallocDefault that property hangs on the
// HIP_HOST_COHERENT environment variable, and non-coherent pinned memory shows the host a kernel's writes only at its end.
#define APE_PINNED (hipHostMallocCoherent | hipHostMallocMapped)
static inline double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int ape_streams_frame_host(ape_streams_t* b, int32_t kind, const float* rows_host, uint32_t flags, void* out_host,
                           int32_t out_dtype, void* stream) {
    if (!b || !rows_host || !out_host) return fail(APE_ERR_INVALID_ARG, "streams_frame_host: NULL argument");
    if (out_dtype != APE_F32 && out_dtype != APE_F64) return fail(APE_ERR_INVALID_ARG, "streams_frame_host: unknown dtype selector");
    if (flags & ~(uint32_t)APE_FLAG_NORMALIZE_INPUT) return fail(APE_ERR_INVALID_ARG, "streams_frame_host: only NORMALIZE_INPUT is accepted");
    int width, I;
    if (!parse_kind_dims(kind & ~APE_PARSE_BIG_ENDIAN, &width, &I)) return fail(APE_ERR_INVALID_ARG, "streams_frame_host: unknown kind %d", kind);
    ape_model* m = b->model;
    HIP_TRY(hipSetDevice(m->dims.device));           // (the consumer thread of an estimator starts on device 0)
    const size_t N = (size_t)b->smooth * b->n_mc;
    const size_t rows_bytes = (size_t)b->S * width * sizeof(float);
    const size_t out_bytes = (size_t)b->S * (25 + 6 * N) * (out_dtype == APE_F64 ? sizeof(double) : sizeof(float));
    if (rows_bytes > b->h_rows_bytes) {
        if (b->h_rows) { HIP_TRY(hipHostFree(b->h_rows)); b->h_rows = nullptr; b->h_rows_bytes = 0; }
        HIP_TRY(hipHostMalloc((void**)&b->h_rows, rows_bytes, APE_PINNED));
        b->h_rows_bytes = rows_bytes;
    }
    if (out_bytes > b->h_out_bytes) {
        if (b->h_out) { HIP_TRY(hipHostFree(b->h_out)); b->h_out = nullptr; b->h_out_bytes = 0; }
        HIP_TRY(hipHostMalloc(&b->h_out, out_bytes, APE_PINNED));
        b->h_out_bytes = out_bytes;
    }
    if (!b->h_status) {
        HIP_TRY(hipHostMalloc((void**)&b->h_status, 64, APE_PINNED));
        *b->h_status = 0u;
    }
    // up to 64 streams: a word per stream that the post kernel's workgroup writes behind its outputs -- the host takes the frame when
    // all are there instead of waiting for the stream's completion signal to travel (a larger bank waits for the stream)
    if (!b->h_done && b->S <= 64) {
        HIP_TRY(hipHostMalloc((void**)&b->h_done, 64 * sizeof(unsigned), APE_PINNED));
        memset(b->h_done, 0, 64 * sizeof(unsigned));
    }
    b->h_done_val += 1;
    if (b->h_done_val == 0) b->h_done_val = 1;
    const double t_begin = now_us();
    memcpy(b->h_rows, rows_host, rows_bytes);
    // (a sentinel, not zero: in the single-launch frame an aborted launch never reaches the tail that writes the word)
    *b->h_status = m->cluster_ok ? 0xFFFFFFFFu : 0u;       // (no cooperative kernel on this model: nothing writes the word, nothing can abort)
    // where the bank steps on the Monte-Carlo latency kernel its extra workgroups build the rows' features (two launches per frame);
    // else the feature builder's own launch in front of the step
    const bool one_launch = bank_on_mc_small(b) && b->inj_masks == nullptr && mc_small_builder_fits(m, b->S, b->n_mc, b->T) && I == m->dims.input_size && b->xring != nullptr;
    float* slot_out; int slot_rep; size_t slot_rep_stride;
    next_slot(b, (size_t)I, &slot_out, &slot_rep, &slot_rep_stride);
    if (one_launch) {
        McFrameParts fr{};
        fr.raw_rows = b->h_rows; fr.raw_width = width; fr.raw_kind = kind & ~APE_PARSE_BIG_ENDIAN; fr.raw_big_endian = (kind & APE_PARSE_BIG_ENDIAN) ? 1 : 0;
        fr.cold = b->frames == 0 ? 1 : 0;
        fr.ring_out = slot_out; fr.ring_stream_stride = (size_t)b->n_mc * b->T * I; fr.ring_rep_stride = slot_rep_stride; fr.ring_rep = slot_rep;
        ++b->frames;                                   // (what ape_streams_push_rows does behind its launch)
        if (int rc = streams_step_impl(b, flags | APE_FLAG_PACKED_MSG, b->h_out, nullptr, out_dtype, stream, b->h_status, &fr)) { --b->frames; return rc; }
    } else {
        if (int rc = ape_streams_push_rows(b, kind, b->h_rows, stream)) return rc;
        if (int rc = streams_step_impl(b, flags | APE_FLAG_PACKED_MSG, b->h_out, nullptr, out_dtype, stream, b->h_status)) return rc;
    }
    const double t_launched = now_us();
    bool seen = false, recovered = false;
    if (b->h_done) {
        // ~50 ms of looking (a frame is tens of microseconds; a launch that gives up takes seconds: the stream wait below covers it)
        volatile unsigned* dw = b->h_done;
        for (long spin = 0; spin < 20000000L && !seen; ++spin) {
            seen = true;
            for (int k = 0; k < b->S; ++k) seen = seen && dw[k] == b->h_done_val;
            if (!seen) __builtin_ia32_pause();
        }
        std::atomic_thread_fence(std::memory_order_acquire);
    }
    const bool fell_through = !seen && b->h_done != nullptr;
    if (!seen) HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    if (*(volatile unsigned*)b->h_status != 0u) {
        recovered = true;
        HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
        // the regressor launch gave up (bounded spins): the step is still the bank's newest one, so it can be issued again on
        // the kernels that need no co-residency -- into the same pinned rows.  (Single-launch frame: the builder workgroup may not
        // have run; its row goes into the ring by the feature builder's own kernel first -- the same slot, the same values.)
        if (one_launch) {
            hipError_t e = ape_launch_parse_rows(b->h_rows, b->S, width, kind & ~APE_PARSE_BIG_ENDIAN, slot_out, APE_F32, I, (size_t)b->n_mc * b->T * I,
                                                 slot_rep, slot_rep_stride, (kind & APE_PARSE_BIG_ENDIAN) ? 1 : 0, (hipStream_t)stream);
            if (e != hipSuccess) return fail(APE_ERR_HIP, "streams_frame_host: feature builder launch failed: %s", hipGetErrorString(e));
        }
        if (int rc = ape_model_recover(m)) return rc;
    } else {
        // the status word was read behind the step's kernels on the step's own stream and one handle serialises on one
        // stream: nothing aborted since the last check, what the journal holds is done
        journal_clear(m);
    }
    const double t_there = now_us();
    memcpy(out_host, b->h_out, out_bytes);
    const double t_end = now_us();
    if (b->fs_trace.empty()) b->fs_trace.resize(4096 * 3, 0.0f);
    float* tr = b->fs_trace.data() + (b->fs_frames % 4096) * 3;
    tr[0] = (float)(t_launched - t_begin); tr[1] = (float)(t_there - t_launched); tr[2] = (float)(t_end - t_there);
    b->fs_frames += 1;
    b->fs_fallback += fell_through ? 1 : 0;
    b->fs_recovered += recovered ? 1 : 0;
    return APE_OK;
}

int ape_streams_frame_stats(ape_streams_t* b, ape_frame_stats_t* out, float* trace_us, int32_t capacity_frames, int32_t* n_out, int32_t reset) {
    if (!b || !out) return fail(APE_ERR_INVALID_ARG, "streams_frame_stats: NULL argument");
    out->frames = b->fs_frames; out->fallback_syncs = b->fs_fallback; out->recovered = b->fs_recovered;
    int32_t n = 0;
    if (trace_us && capacity_frames > 0 && !b->fs_trace.empty()) {
        const uint64_t have = b->fs_frames < 4096 ? b->fs_frames : 4096;
        n = (int32_t)(have < (uint64_t)capacity_frames ? have : (uint64_t)capacity_frames);
        for (int32_t i = 0; i < n; ++i) {
            const uint64_t frame = b->fs_frames - (uint64_t)n + (uint64_t)i;
            memcpy(trace_us + 3 * (size_t)i, b->fs_trace.data() + (frame % 4096) * 3, 3 * sizeof(float));
        }
    }
    if (n_out) *n_out = n;
    if (reset) { b->fs_frames = b->fs_fallback = b->fs_recovered = 0; }
    return APE_OK;
}

int ape_streams_profile(ape_streams_t* b, int32_t enable) {
    if (!b) return fail(APE_ERR_INVALID_ARG, "streams_profile: NULL bank");
    HIP_TRY(hipSetDevice(b->model->dims.device));
    if (enable && b->prof_ev.empty()) {
        b->prof_ev.resize(512, nullptr);
        for (auto& ev : b->prof_ev) HIP_TRY(hipEventCreate(&ev));
    }
    b->prof_on = enable != 0;
    b->prof_n = 0;
    return APE_OK;
}

int ape_streams_profile_read(ape_streams_t* b, double* kernel_ms_sum, int32_t* launches) {
    if (!b || !kernel_ms_sum || !launches) return fail(APE_ERR_INVALID_ARG, "streams_profile_read: NULL argument");
    double sum = 0.0;
    for (int i = 0; i < b->prof_n; ++i) {
        HIP_TRY(hipEventSynchronize(b->prof_ev[2 * i + 1]));
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, b->prof_ev[2 * i], b->prof_ev[2 * i + 1]));
        sum += ms;
    }
    *kernel_ms_sum = sum; *launches = b->prof_n;
    b->prof_n = 0;
    return APE_OK;
}

// one journaled call again, on the kernels that need no co-residency (m->replaying is set)
static int replay_entry(ape_model_t* m, const ApeJournalEntry& e) {
    switch (e.kind) {
        case ApeJournalEntry::FORWARD:
            return ape_lstm_forward(m, (const float*)e.in0, e.B, e.T, e.flags, (const float*)e.in1, e.dropout_p, e.seed, (float*)e.out0, e.stream);
        case ApeJournalEntry::FORWARD_HS:
            return ape_lstm_forward_hs(m, (const float*)e.in0, e.B, e.T, e.flags, nullptr, e.dropout_p, e.seed, (const float*)e.in1,
                                       (const float*)e.in2, (float*)e.out0, e.stream);
        case ApeJournalEntry::FK:
            return ape_fk(m, e.in0, e.i0, e.B, e.i1, e.out0, e.i2, e.stream);
        case ApeJournalEntry::MSG:
            return ape_msg_reduce(m, (const double*)e.in0, e.B, (double*)e.out0, e.stream);
        case ApeJournalEntry::INFER:
            return ape_infer(m, (const float*)e.in0, e.B, e.T, e.flags, (float*)e.out0, e.out1, e.i2, e.stream);
        case ApeJournalEntry::STEP: {
            ape_streams* b = e.bank;
            b->steps = e.bank_steps; b->mc_calls = e.bank_mc_calls;         // (frames unchanged: checked by the caller)
            return ape_streams_step(b, e.flags, e.out0, e.out1, e.i2, e.stream);
        }
    }
    return fail(APE_ERR_INVALID_ARG, "recover: unknown journal entry");
}

// internal diagnostic accessor (not part of the public header): copies the 2 stamp words
int ape_debug_read_stamps(ape_model_t* m, unsigned long long out[14]) {
    if (!m || !m->cluster_ok) return APE_ERR_INVALID_ARG;
    HIP_TRY(hipMemcpy(out, m->xflags + m->xflag_bytes / sizeof(unsigned) + 4, 112, hipMemcpyDeviceToHost));
    return APE_OK;
}

int ape_debug_read_wg(ape_model_t* m, unsigned long long out[256 * 8]) {
    if (!m || !m->dbg_wg) return APE_ERR_INVALID_ARG;
    HIP_TRY(hipMemcpy(out, m->dbg_wg, 256 * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return APE_OK;
}

// internal: read `n` control words of the MLP pipeline from word `first` on (its segment stamps, tests/tools/pipe_stamps.py)
int ape_debug_peek_pipe(ape_model_t* m, int first, unsigned* out, int n) {
    if (!m || !m->ffp_ok || !out || first < 0 || n < 1 || (size_t)(first + n) > m->ffp_ctl_words) return APE_ERR_INVALID_ARG;
    HIP_TRY(hipSetDevice(m->dims.device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, m->ffp_ctl + first, (size_t)n * sizeof(unsigned), hipMemcpyDeviceToHost));
    return APE_OK;
}

// internal: the batch split of lstm_forward for a device with `n_cus` CUs, no GPU needed.
// out = {rows to the batch-tile kernel, row tiles per cluster, clusters per launch, cluster launches, cluster capacity}
// `c32`: the model / call is eligible for the second-generation f32 kernel (what lstm_forward_impl passes for an eval-mode,
// last-step call on a 2 x 256 model); out[5] = the kernel that serves the rest (PLAN_*), with the second-generation kernel out[1]
// (row tiles) is 2 = its 32-row clusters
int ape_debug_plan2(const ape_dims_t* dims, int n_cus, int B, int T, int cdrop, int c32, int out[6]) {
    if (!dims || !out || n_cus < 1 || B < 1 || T < 1) return APE_ERR_INVALID_ARG;
    const int H = dims->hidden_size;
    const int cap = cluster_capacity(n_cus, H);
    out[4] = cap; out[5] = PLAN_NONE;
    if (cap < 1) { out[0] = B; out[1] = out[2] = out[3] = 0; return APE_OK; }
    const bool g2 = c32 != 0 && !cdrop && f16v2_capacity(n_cus) > 0;
    const int gen2 = !g2 ? 0 : ape_cluster32_supported(H, dims->num_layers, padded_input(dims->input_size)) ? 32
                     : ape_cluster16_supported(H, dims->num_layers, padded_input(dims->input_size))
                           ? ((ape_level16_supported(H, dims->num_layers, padded_input(dims->input_size)) && ape_level16_max_clusters(n_cus) > 0) ? 48 : 16) : 0;
    long long n16 = 0;
    if (B > 4) n16 = (long long)tile16_wave_rows(n_cus) * auto_tile16_waves(dims, n_cus, B, T, cdrop != 0, gen2);
    if (n16 > B) n16 = B;
    out[0] = (int)n16;
    const int rest = B - (int)n16;
    out[1] = out[2] = out[3] = 0;
    if (rest > 0) {
        out[5] = (B <= 4 && !cdrop) ? PLAN_SMALL : rest_kernel(rest, T, gen2, n_cus);
        if (out[5] == PLAN_C32 || out[5] == PLAN_C16) {
            const int rpl2 = 32 * f16v2_capacity(n_cus);
            out[1] = 2;
            out[3] = (rest + rpl2 - 1) / rpl2;
            out[2] = ((rest < rpl2 ? rest : rpl2) + 31) / 32;
        } else if (out[5] == PLAN_LV16) {
            const int rpl3 = 32 * ape_level16_max_clusters(n_cus);
            const bool single = rest <= rpl3 / 2;                 // one row tile per cluster (APE_FLAG_LV16_SINGLE)
            out[1] = single ? 1 : 2;
            out[3] = (rest + rpl3 - 1) / rpl3;
            out[2] = single ? (rest + 15) / 16 : ((rest < rpl3 ? rest : rpl3) + 31) / 32;
        } else {
            const int nmt = cluster_nmt(n_cus, H, rest, cdrop != 0), rpl = 16 * nmt * cap;
            out[1] = nmt;
            out[3] = (rest + rpl - 1) / rpl;
            const int first = rest < rpl ? rest : rpl;
            out[2] = (first + 16 * nmt - 1) / (16 * nmt);
        }
    }
    return APE_OK;
}
int ape_debug_bank_chunks(int up128, int S, int T, int n_mc, long long out[3]) {
    if (!out || S < 1 || T < 1 || n_mc < 1) return APE_ERR_INVALID_ARG;
    long long chunk = 0;
    out[0] = bank_chunk_plan(up128 != 0, S, T, n_mc, &chunk) ? 1 : 0;
    out[1] = chunk;
    out[2] = out[0] ? ((long long)S * n_mc + chunk - 1) / chunk : 0;
    return APE_OK;
}

// the route ape_streams_set_mc plans for a Monte-Carlo bank of S streams x n_mc samples on a healthy model under APE_KERNEL_AUTO, float32,
// dropout on (pure host arithmetic, for the CPU tests): out = {layer 0 shared 0/1, route BANK_*, launch A's kernel BANK_A_*, chunk rows}
int ape_debug_bank_route(const ape_dims_t* dims, int n_cus, int S, int T, int n_mc, long long out[4]) {
    if (!dims || !out || n_cus < 1 || S < 1 || T < 1 || n_mc < 1) return APE_ERR_INVALID_ARG;
    const int H = dims->hidden_size, L = dims->num_layers, O = dims->output_size, KX = padded_input(dims->input_size);
    const bool lstm = dims->model_kind == APE_MODEL_LSTM;
    const bool cluster_ok = lstm && ape_cluster_supported(H, L, KX) && cluster_capacity(n_cus, H) >= 1;
    const bool upper_ok = lstm && ((H == 256 && L == 2) || (H == 128 && L == 3));
    const bool up32_ok = cluster_ok && ape_cluster32_supported(H, L, KX) && f16v2_capacity(n_cus) > 0 && ape_upper32_supported(H, L, O);
    const bool up128_ok = cluster_ok && ape_upper128_supported(H, L, O) && (n_cus / 4) / 8 * 8 >= 8;
    const bool can_up128 = up128_ok, can_up32 = (up32_ok && f16v2_capacity(n_cus) >= 8) || can_up128;
    const long long rows = (long long)S * n_mc;
    const bool shared = upper_ok && n_mc >= 2 && bank_shares_layer0(rows, n_cus, can_up32, can_up128);
    long long chunk = 0;
    int route = BANK_FUSED;
    if (shared) route = (can_up32 && bank_chunk_plan(can_up128, S, T, n_mc, &chunk)) ? (can_up128 ? BANK_UPPER128 : BANK_UPPER32) : BANK_SHARED_TILE16;
    out[0] = shared ? 1 : 0; out[1] = route; out[2] = bank_launch_a_form(route, H, KX, n_cus, S, cluster_ok); out[3] = chunk;
    return APE_OK;
}

int ape_debug_plan(const ape_dims_t* dims, int n_cus, int B, int T, int cdrop, int out[5]) {
    int o6[6];
    const int rc = ape_debug_plan2(dims, n_cus, B, T, cdrop, 0, o6);
    if (rc == APE_OK) for (int i = 0; i < 5; ++i) out[i] = o6[i];
    return rc;
}

const char* ape_model_last_kernel(const ape_model_t* m) { return m ? m->last_kernel : ""; }

const char* ape_lstm_kernel_name(const ape_model_t* m, int32_t B, int32_t T) {
    if (!m) return "";
    if (m->dims.model_kind == APE_MODEL_FF)        // (B rows of the last step, eval mode)
        return (m->ffp_ok && m->ffp_on && B >= 64 * m->n_cus) ? "ape_mlp_pipe" : m->kernel_name.c_str();
    if (m->precision == APE_PRECISION_F16)
        return (m->f16_v2 && ape_cluster_f16v2_supported(m->dims.hidden_size, m->dims.num_layers, m->KX) &&
                f16v2_capacity(m->n_cus) > 0 && B > 256) ? "ape_lstm_cluster_f16v2" : "ape_lstm_cluster_f16";
    if (!m->cluster_ok || m->kernel_choice == APE_KERNEL_TILE16) return m->kernel_name.c_str();
    // ImuPoseLSTM above 512 windows: one layer per launch on lstm_upper32.hip's persistent clusters (lstm_forward_impl)
    if (m->split32_ok && B > 512 && m->kernel_choice == APE_KERNEL_AUTO && m->c32_on && m->precision == APE_PRECISION_F32)
        return "ape_lstm_upper32<32, true> + <32, false>";
    // under AUTO the kernel that takes the larger part of an eval-mode batch of this shape
    if (m->kernel_choice == APE_KERNEL_AUTO && B > 4 && T >= 1) {
        if (2LL * tile16_wave_rows(m->n_cus) * auto_tile16_waves(&m->dims, m->n_cus, B, T, false, gen2_of(m), m->wide_cluster) > B) return m->kernel_name.c_str();
    }
    if (m->c32_ok && m->c32_on && m->precision == APE_PRECISION_F32 && B > 512)      // (two instantiations: lstm_cluster32.hip on ENDS)
        return T <= APE_C32_ENDS_MAX_T ? "ape_lstm_cluster32<256, 2, 32, true>" : "ape_lstm_cluster32<256, 2, 32, false>";
    if (m->c16_ok && m->c32_on && m->precision == APE_PRECISION_F32) {
        const int k = rest_kernel(B, T, gen2_of(m), m->n_cus);
        if (k == PLAN_LV16) return "ape_lstm_level16<128, 3, 64>";
        if (k == PLAN_C16) return "ape_lstm_cluster16<128, 3, 64, 2>";
    }
    return m->cluster_name.c_str();
}

}  // extern "C"
