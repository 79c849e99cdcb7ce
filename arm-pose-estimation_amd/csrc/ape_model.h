// The model and stream-bank handles behind the C ABI: shared by ape_api.hip and by the test-hooks translation unit
// (ape_debug.hip, linked into lib/diag/libape_hip_testhooks.so only).
#pragma once
#include <string>
#include <vector>

#include "../../include/ape_hip.h"
#include "ape_internal.h"

struct ape_streams;

// One compute call on a model since its last successful check, kept so that ape_model_recover can re-issue it on the
// kernels that need no co-residency when a weight-stationary launch gave up (include/ape_hip.h, "Aborted launches").
struct ApeJournalEntry {
    enum Kind { FORWARD, FORWARD_HS, FK, MSG, INFER, STEP } kind;
    const void* in0;                 // x / preds / est
    const void* in1;                 // masks / h0
    const void* in2;                 // c0
    void* out0;                      // y / est / msg
    void* out1;                      // est (infer) / tail (step)
    int32_t B, T, i0, i1, i2;        // sizes and dtype selectors
    uint32_t flags;
    float dropout_p;
    uint64_t seed;
    void* stream;
    ape_streams* bank;               // STEP: the bank and its counters in front of the step
    long long bank_frames, bank_steps;
    unsigned long long bank_mc_calls;
};
#define APE_JOURNAL_CAP 64

struct ape_model {
    ape_dims_t dims{};
    void* slab = nullptr;          // the one device allocation every fixed-size buffer below points into
    size_t slab_bytes = 0;
    int n_cus = 0;                 // hipDeviceProp_t::multiProcessorCount of the model's device (256 on a whole MI355X)
    int KX = 0;                    // LSTM layer-0 input width, padded to the kernels' k-blocking
    int lstm_in = 0;               // LSTM layer-0 input width (input_size; 256 behind ImuPoseLSTM's input layer)
    int KXpre = 0;                 // ImuPoseLSTM: padded width of the input layer's input
    float* z_ws = nullptr;         // ImuPoseLSTM: [cap rows, 256] activations of the input layer
    float* hseq_ws = nullptr;      // all-steps mode of the cluster kernel: [cap rows = B*T, H] top-layer outputs
    size_t hseq_cap = 0;
    size_t z_cap = 0;
    f32x4* wpack[APE_MAX_LAYERS] = {nullptr, nullptr, nullptr};
    float* bias[APE_MAX_LAYERS] = {nullptr, nullptr, nullptr};
    float* w_out = nullptr;
    float* b_out = nullptr;
    double* stats = nullptr;       // device: xx_m[I] xx_s[I] yy_m[O] yy_s[O] 1/xx_s[I]
    bool has_weights = false, has_stats = false;
    double body[9];
    float* y_ws = nullptr;         // [cap, O] intermediate of ape_infer
    int y_cap = 0;
    // weight-stationary cluster kernel (lstm_cluster.hip)
    bool cluster_ok = false;
    int kernel_choice = APE_KERNEL_AUTO;
    float* wcl[APE_MAX_LAYERS] = {nullptr, nullptr, nullptr};
    void* wcl16[APE_MAX_LAYERS] = {nullptr, nullptr, nullptr};   // binary16 fragments of the fp16 variant
    float* wcl32[APE_MAX_LAYERS] = {nullptr, nullptr, nullptr};  // 32x32x2 fragments of the second-generation f32 cluster kernel
    float* wcls[APE_MAX_LAYERS] = {nullptr, nullptr, nullptr};   // the latency kernel's H/8-member form (two units per wave)
    char* hxs = nullptr;             // latency kernel: [256 B: launch number][granules {h, tag}: layer, parity, 4 rows, H units]
    size_t hxs_bytes = 0;
    float* wmc[APE_MAX_LAYERS] = {nullptr, nullptr, nullptr};    // Monte-Carlo latency kernel: layers >= 1, wave = K quarter (4x4x1 MFMA fragments)
    float* wmc16[APE_MAX_LAYERS] = {nullptr, nullptr, nullptr};  // ... and as 16 x 16 x 4 fragments (16-row clusters: wave = column tile x K half)
    char* gxm = nullptr;             // ... [256 B: launch number][8 clusters of granules]
    size_t gxm_cluster_bytes = 0;
    bool mcs_ok = false;             // lstm_mc_small.hip covers this model on this device
    int small_uw = 4;                // hidden units per wave of the latency kernel: 2 when one XCD holds H/8 members
    bool c32_ok = false;            // lstm_cluster32.hip covers this model (2 x 256) on this device
    bool c32_on = true;             // ... and is not switched off (APE_KERNEL_CLUSTER_GEN1)
    bool gen1_classes = true;       // first-generation f32 kernel: XCD-class cluster formation where the grid allows it
    bool c16_ok = false;            // lstm_cluster16.hip covers this model (3 x 128) on this device (switched with c32_on)
    bool lv16_ok = false;           // ... and lstm_level16.hip its short windows (switched with c32_on too)
    char* gx16 = nullptr;           // ... [256 B: launch number][granules {h, tag}: cluster, row tile, parity, layer]
    size_t gx16_bytes = 0;
    int precision = APE_PRECISION_F32;
    bool wide_cluster = false;      // ImuPoseLSTM: the f32 first-generation cluster kernel with a 256-wide layer-0 input, nothing else
    bool small_batch_path = true;   // B <= 4 on the VALU/shuffle variant of the cluster kernel
    bool f16_v2 = true;             // fp16 precision: batches > 256 rows on the row-set-pipelined kernel (lstm_cluster_f16v2.hip)
    bool upper_ok = false;          // layers 1.. can run on their own over a shared layer-0 sequence (stream bank, MC mode)
    bool up32_ok = false;           // ... and on the weight-stationary upper-layer kernel (lstm_upper32.hip: 2 x 256 models)
    bool up128_ok = false;          // ... or on lstm_upper128.hip (the 3 x 128 model: layers 1 and 2 in four-member clusters)
    bool split32_ok = false;        // ImuPoseLSTM: the 2 x 256 LSTM behind the input layer, one layer per launch on lstm_upper32.hip's persistent clusters
    float* zfrag_ws = nullptr;      // ... its workspaces: the input layer's activations in fragment order [tiles][T][32 KB],
    float* hfrag_ws = nullptr;      //     layer 0's output sequence (same shape, layer 1's input),
    float* ypart_ws = nullptr;      //     head partial sums [tiles * 32][8][16]
    size_t split_tiles_cap = 0, split_steps_cap = 0;      // tiles x steps the first two hold
    float* wup128[APE_MAX_LAYERS] = {nullptr, nullptr, nullptr};   // its register image of layers 1, 2
    float* hx = nullptr;           // exchange slices
    size_t hx_bytes = 0;
    unsigned long long* dbg_wg = nullptr;   // 256 x 8 words, written by diagnostic builds of the cluster kernel only
    unsigned* xcc_slots = nullptr; // small-batch kernel: 64 words its members publish their XCD in (zero between launches)
    unsigned* xflags = nullptr;    // [flag words..., status word]
    size_t xflag_bytes = 0;        // bytes of the flag block (multiple of 16), status word follows
    // MLP regressor (APE_MODEL_FF)
    float *ffp_wa0 = nullptr, *ffp_wa1 = nullptr, *ffp_wb2 = nullptr, *ffp_wbo = nullptr;   // mlp_pipe.hip: the two stages' register files
    float* ffp_ring = nullptr;       // ... its ring of tiles between the stages
    size_t ffp_ring_bytes = 0;
    unsigned* ffp_ctl = nullptr;     // ... class tickets, status, departure counter, per-pair full / empty words (zero between launches)
    size_t ffp_ctl_words = 0;
    bool ffp_ok = false, ffp_on = true;
    f32x4* ff_wpack[APE_MAX_FF_LAYERS] = {};
    float* ff_bias[APE_MAX_FF_LAYERS] = {};
    std::string kernel_name, cluster_name;
    const char* last_kernel = "";   // the LSTM kernel the newest compute call launched last (ape_model_last_kernel)
    // calls since the last successful check (ape_model_recover replays them), and what became of aborted launches
    ApeJournalEntry journal[APE_JOURNAL_CAP];
    int journal_n = 0;
    bool journal_overflow = false;
    bool replaying = false;         // ape_model_recover is re-issuing: no journaling, no cooperative kernel
    ape_model_stats_t stats_counts{};
};

struct ape_streams {
    ape_model* model = nullptr;
    int S = 0, T = 0, smooth = 0;
    int n_mc = 1;                // Monte-Carlo samples per stream and step
    bool mc = false;             // dropout on (ape_streams_set_mc was called)
    bool shared_l0 = false;      // MC mode with layer 0 computed once per stream (two launches per step)
    float dropout_p = 0.0f;
    unsigned long long seed = 0, mc_calls = 0;
    float* xring = nullptr;      // [S,n_mc,T,I] feature rows, slot = frame mod T (a stream's n_mc windows are copies)
    float* yring = nullptr;      // [S,smooth,n_mc,O] model outputs, slot = step mod smooth
    float* y_new = nullptr;      // [S,n_mc,O]
    double* post_part = nullptr; // split post-filter (stream_post_device.h): [S][chunks][21] partial sums + [S] tickets behind them, or
    unsigned* post_cnt = nullptr; //   nullptr where every stream keeps a workgroup of its own (many streams, or stacks of <= 64 rows)
    // shared-layer-0 route on the weight-stationary upper-layer kernel (lstm_upper32.hip): the sample rows go through it in
    // chunks of `chunk_rows` (a multiple of 32), each expand -> LSTM -> head reduce over the two workspaces below
    bool up32 = false;
    bool up128 = false;          // the same route for the 3 x 128 model (lstm_upper128.hip; launch A stays on the batch-tile kernel)
    unsigned* maskbits = nullptr;   // ... keep bits of layer 1's outputs [chunk tiles][T][128]
    int chunk_rows = 0;
    float* xfrag = nullptr;      // [chunk tiles][T][32 KB] masked layer-0 output in MFMA fragment order
    float* ypart = nullptr;      // [chunk rows][8][16] head partial sums
    float* xfrag0 = nullptr;     // [S / 32][T][4 KB] layer 0's input tiles, fragment order
    float* hfrag = nullptr;      // [S / 32][T][32 KB] layer 0's output sequence, fragment order (launch B's input builder reads it)
    bool prof_on = false;        // ape_streams_profile: event pairs around the dominant kernel's launches
    int prof_n = 0;
    std::vector<hipEvent_t> prof_ev;
    const float* inj_masks = nullptr;   // test hooks only (ape_debug_set_bank_masks): injected multipliers [L-1, S*n_mc, T, H] instead of Philox
    long long frames = 0;        // rows pushed since the last reset
    long long steps = 0;         // predictions made since the last reset
    // host frames (ape_streams_frame_host): pinned, device-visible staging the kernels read the raw rows from and write the
    // datagram rows (+ the model's status word) to -- no copy command in the frame, one stream synchronisation
    float* h_rows = nullptr;     // [S, 55|28]
    void* h_out = nullptr;       // [S, 25 + 6 * smooth * n_mc] of the frame's dtype
    unsigned* h_status = nullptr;
    unsigned* h_done = nullptr;  // [64] pinned: the post kernel's per-stream "outputs are there" words of a host frame
    unsigned h_done_val = 0;     // ... and the value this frame's are awaited with
    size_t h_rows_bytes = 0, h_out_bytes = 0;
    // where a host frame's time goes (ape_streams_frame_stats): a ring of the last 4096 frames' {launch, wait, copy} microseconds
    std::vector<float> fs_trace;
    uint64_t fs_frames = 0, fs_fallback = 0, fs_recovered = 0;
};
