// Internal declarations shared by the C-ABI (ape_api.hip) and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/ape_hip.h"

#define APE_MAX_LAYERS 3
#define APE_MAX_INPUT 64
#define APE_MAX_OUTPUT 32
#define APE_TILE_ROWS 16          // windows per workgroup in the batch-tile LSTM kernel
#define APE_XCC_WORDS 1024
#define APE_C32_ENDS_MAX_T 8      // lstm_cluster32.hip: windows of up to this many steps run the instantiation with the end forms
#define APE_LDS_BYTES (160 * 1024) // LDS of a gfx950 CU: the dynamic-LDS limit every kernel instantiation is raised to, once
// internal timing-only ablation switches: the diagnostic builds alone (make diag / make ablate) read them, outputs are wrong.  Their bit
// values overlap with NO public flag or selector of include/ape_hip.h (round 5: two of them used to).
#define APE_DIAG_NO_EXCHANGE 0x40000000u
#define APE_DIAG_NO_ACT      0x20000000u
#define APE_DIAG_STAMP       0x10000000u
#define APE_DIAG_NO_MFMA     0x04000000u   // fp16 v2 / upper-layer kernel: skip the matrix work
#define APE_DIAG_NO_XSTAGE   0x00200000u   // ... skip the staging / fetch of the next step's inputs
#define APE_DIAG_NO_BARRIER  0x00100000u   // upper-layer kernel: no workgroup barrier at a section's top

#define APE_FLAG_XCD_CLASSES 0x00800000u   // internal (set by the launcher): the first-generation kernel forms its clusters within block-index classes
#define APE_FLAG_LV16_SINGLE 0x00800000u   // internal (set by the launcher, lstm_level16.hip only -- the same bit means nothing else there): one row tile per cluster

// Hand-over form of a flag-based kernel, in one place: plain (write-back) payload stores ONLY when the caller opted in with
// APE_FLAG_IN_XCD_PLAIN AND the cluster verified at run time that all its members share an XCD (`same_xcd`); otherwise write-through.
#define APE_HANDOVER_IN_L2(flags, same_xcd) ((same_xcd) && ((flags) & APE_FLAG_IN_XCD_PLAIN) != 0u && ((flags) & APE_FLAG_ANY_PLACEMENT) == 0u)

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Kernel arguments of the batch-tile LSTM kernel (passed by value).
struct LstmParams {
    const float* x;                     // [B,T,I]
    float* y;                           // [B,O] or [B,T,O]
    const f32x4* wpack[APE_MAX_LAYERS]; // per layer: MFMA-fragment-ordered [W_ih | W_hh]
    const float* bias[APE_MAX_LAYERS];  // per layer [4H] = b_ih + b_hh
    const float* w_out;                 // [O,H]
    const float* b_out;                 // [O]
    const double* xx_m;                 // [I] (device) or nullptr
    const double* xx_s;                 // [I]
    const float* masks;                 // [L-1,B,T,H] or nullptr
    int B, T, I, O;
    int KX;                             // input width padded to a multiple of 32
    int x_ring;                         // time step t lives in slot (t + x_ring) mod T of a window (0: linear)
    unsigned flags;
    float dropout_p;
    unsigned long long seed;
    size_t x_row_stride;                // floats between the windows of consecutive rows (0: T*I, a dense [B,T,I])
    float* hseq;                        // [B,T,H]: every step's top-layer output goes here too (or nullptr)
    int x_group;                        // wide-input instantiations: row b reads window b / x_group of x, and with
                                        // DROPOUT_PHILOX the input is masked as the output of a layer 0 would be
                                        // (0: row b reads window b, no mask)
    int layer_base;                     // model layer that this launch's layer 0 is (a launch over the UPPER layers of a
                                        // model keeps the model's layer numbers in its Philox counters)
    const float* h0;                    // [L,hs_rows,H] initial hidden state (DropoutLSTM.forward(x, hs), nn_models.py:180-189)
    const float* c0;                    // [L,hs_rows,H] initial cell state; both nullptr: zeros
    int hs_rows;                        // batch size of h0 / c0
};

// Kernel arguments of the weight-stationary cluster LSTM kernel.
struct ClusterParams {
    const float* x;                     // [B,T,I]
    float* y;                           // [B,O]
    const float* wcl[APE_MAX_LAYERS];   // per layer [member GH][wave 4][NW_l][lane 64]: B-fragment registers
    const float* bias[APE_MAX_LAYERS];  // per layer [4H] = b_ih + b_hh
    const float* w_out;                 // [O,H]
    const float* b_out;                 // [O]
    const double* xx_m;
    const double* xx_s;
    const double* xx_r;                 // [I] 1 / xx_s, rounded once on the host
    float* hx;                          // exchange slices [cluster][L][parity 2][member GH][wave 4][row MR][4 units] (f32 cluster kernel)
    size_t hx_bytes;
    unsigned* xflags;                   // [cluster][L][GH] epoch flags, zeroed before every launch
    unsigned* ticket;                   // [1] arrival counter (re-zeroed by the last workgroup out)
    unsigned* done;                     // [1] departure counter
    unsigned* status;                   // [1] sticky: 1 = a bounded spin gave up
    const float* masks;                 // [L-1,B,T,H] injected dropout masks or nullptr
    int B, T, I, O;
    int x_ring;                         // time step t lives in slot (t + x_ring) mod T of a window (0: linear)
    unsigned flags;
    float dropout_p;
    unsigned long long seed;
    float* hseq;                        // [B,T,H] every step's top-layer output (all-steps mode), or nullptr
    size_t x_row_stride;                // first-generation f32 kernel: floats between consecutive rows of x (0: T * I; a Monte-Carlo bank's ring
                                        // keeps n_mc window slots per stream and layer 0 reads the first of each)
    int row_base;                       // first-generation kernel, Philox dropout: global index of this launch's row 0 in the caller's batch
                                        // (the counters name GLOBAL rows: the samples of a call do not depend on how it is split over launches)
    unsigned long long* dbg_wg;         // diagnostic builds only: 8 words per workgroup (ticket, XCC, clocks)
    unsigned* seq;                      // latency kernel: [1] launch number of this model (the upper bits of its granule tags)
    // latency kernel only: the post-filter in the same launch (ape_infer at B <= 4).  est row b = FK of y row b, de-normalised when
    // fk_yy_m is set; nullptr: no tail
    void* fk_est;                       // [B,W] of fk_est_dtype
    const double* fk_yy_m;
    const double* fk_yy_s;
    double fk_body[9];                  // larm_vec, uarm_vec, uarm_orig_rh
    int fk_layout, fk_W, fk_est_dtype;
    unsigned* xcc_slots;                // APE_XCC_WORDS words, zero between launches.  [0,64): small-batch kernel, (0x10 | XCC id) of
                                        // its members; [64,192): fp16 v2 kernel's 8 class tickets, one per 64-byte line;
                                        // [192, ...): its per-workgroup XCD words
};

// Kernel arguments of the weight-stationary upper-layer kernel of the Monte-Carlo stream bank (lstm_upper32.hip): one chunk of
// sample rows, tiles of 32.
struct UpperParams {
    const float* xfrag;                 // [n_tiles][T][k-block 32][window 32][8 units] masked input, MFMA fragment order
    size_t xfrag_bytes;                 // < 4 GiB (one buffer descriptor)
    float* ypart;                       // [n_tiles * 32][member 8][16] head partial sums
    const float* w;                     // [member 8][wave 4][64][lane 64][4]: 256 weight registers per lane ([W_ih | W_hh], wcl32 layout)
    const float* bias;                  // [4H] = b_ih + b_hh
    const float* w_out;                 // [O,H]
    float* hx;                          // exchange slices [cluster][set 2][parity 2][member 8][wave 4][window 32][8 units]
    size_t hx_bytes;
    unsigned* xflags;                   // [cluster][set 2][32] epoch flags, zero between launches
    unsigned* done;                     // [1] departure counter
    unsigned* status;                   // [1] sticky: 1 = a bounded spin gave up
    unsigned* xcc_slots;                // as ClusterParams::xcc_slots (class tickets + per-workgroup XCD words)
    unsigned long long* dbg_wg;         // diagnostic builds only
    float* hseq;                        // layer-0 form only: [n_tiles][T][32 KB] every step's slices (fragment order)
    size_t hseq_bytes;
    int T, O, n_tiles;
    unsigned flags;                     // APE_FLAG_IN_XCD_PLAIN / APE_FLAG_ANY_PLACEMENT only
};

// Kernel arguments of the weight-stationary kernel for layers 1 and 2 of the 3 x 128 model in a Monte-Carlo stream bank
// (lstm_upper128.hip): one chunk of sample rows, tiles of 32.
struct Upper128Params {
    const float* xfrag;                 // [n_tiles][T][k-block 16][row 32][8 units]: layer 0's output under the rows' masks, fragment order
    size_t xfrag_bytes;
    const unsigned* maskbits;           // [n_tiles][T][unit 128]: bit n = row n of the tile keeps layer 1's output of that unit and step
    float* ypart;                       // [n_tiles * 32][member 4][16] head partial sums
    const float* w[2];                  // layers 1, 2: [member 4][wave 4][32][lane 64][4] = 128 weight registers per lane each
    const float* bias[2];               // [4H] = b_ih + b_hh
    const float* w_out;                 // [O,H]
    float* hx;                          // exchange [cluster][set 2][kind 3: h_1, h_1 masked, h_2][parity 2][16 KB]
    size_t hx_bytes;
    unsigned* xflags;                   // [cluster][set 2][layer 2][16] epoch flags, zero between launches
    unsigned* done;
    unsigned* status;
    unsigned* xcc_slots;
    int T, O, n_tiles;
    unsigned flags;                     // APE_FLAG_IN_XCD_PLAIN / APE_FLAG_ANY_PLACEMENT only
    float dropout_p;
    unsigned long long* dbg_wg;         // diagnostic build only (APE_CLUSTER_STAMPS): shader-clock sums of cluster 0 / member 0 / wave 0
};

// Kernel arguments of the input builder of the layer-0 launch (ape_x_frag_kernel, lstm_upper32.hip).
struct XFragParams {
    const float* x;                     // window rings: stream s at x + s * x_row_stride, [T][I]
    float* xfrag;                       // [tiles][T][4 KB]
    const double* xx_m;                 // [I] or nullptr (no z-score)
    const double* xx_s;
    size_t x_row_stride;
    int S, T, I, x_ring;
};

// Kernel arguments of the input builder of that launch (ape_mc_expand_kernel, lstm_upper32.hip).
struct ExpandParams {
    const float* hseq;                  // [S,T,H] layer-0 output sequence of every stream (launch A)
    float* xfrag;                       // as UpperParams::xfrag
    long long row_base;                 // global index of this chunk's first sample row (a multiple of 32)
    int rows;                           // sample rows in this chunk
    int T, n_mc;
    int layer;                          // model layer whose output hseq is (Philox counter word)
    float dropout_p;
    unsigned long long seed;
    int hseq_frag;                      // 1: hseq is the layer-0 cluster kernel's [S / 32][T][k-block 32][stream 32][8] (fragment order)
    const float* masks;                 // injected multipliers [L-1, masks_rows, T, H] instead of the Philox draws (test hooks only), or nullptr
    long long masks_rows;               // sample rows of the whole bank (the masks' row stride)
};

#define APE_MAX_FF_LAYERS 8          // input layer + up to 7 hidden layers of the MLP regressor

// Kernel arguments of the MLP (DropoutFF) kernel.
struct MlpParams {
    const float* x;                        // rows at x[n * row_stride + row_offset + k]
    float* y;                              // [N,O]
    const f32x4* wpack[APE_MAX_FF_LAYERS]; // per layer MFMA-fragment-ordered weights
    const float* bias[APE_MAX_FF_LAYERS];  // per layer [H]
    const float* w_out;                    // [O,H]
    const float* b_out;                    // [O]
    const double* xx_m;
    const double* xx_s;
    const float* mask;                     // [N,H] dropout mask in front of the output layer, or nullptr
    float* hidden_out;                     // [N,H]: write the last hidden activation here INSTEAD of the output layer
    float neg_slope;                       // 0.01 leaky_relu (DropoutFF), 0 relu (ImuPoseLSTM's input layer)
    size_t row_stride, row_offset;
    int N, I, O, KX, n_hidden;
    unsigned flags;
    float dropout_p;
    unsigned long long seed;
};

struct FkParams {
    const void* preds;   // [N,O] f32 or f64
    void* est;           // [N,W] f32 or f64
    const double* yy_m;  // [O] device, or nullptr (no de-normalisation)
    const double* yy_s;
    double body[9];      // larm_vec, uarm_vec, uarm_orig_rh
    int N, O, W, layout;
};

// Kernel arguments of the per-stream smoothing + message kernel (stream bank).
struct StreamPostParams {
    const float* y_new;  // [S,n_mc,O] NN targets of this step (model output, still normalised)
    float* yring;        // [S,smooth,n_mc,O] the last `smooth` predictions (n_mc samples each) of every stream
    void* msg;           // [S,25] of msg_dtype
    void* tail;          // [S,smooth*n_mc,6] of msg_dtype (hand xyz, elbow xyz of every stacked row) or nullptr
    const double* yy_m;  // [O] or nullptr (no de-normalisation)
    const double* yy_s;
    double body[9];
    int S, O, W, layout, smooth;
    int n_mc;            // Monte-Carlo samples per stream and step (1: deterministic bank)
    int pos;             // ring slot of this step's prediction
    int cold;            // 1: first step after a reset -- every slot takes this prediction (estimator.py:114-115)
    int msg_dtype;
    int packed;          // 1: msg rows are [25 + 6*smooth*n_mc] wide and carry the tail behind the message
    const unsigned* status_in;   // host frames: the model's sticky status word ...
    unsigned* status_out;        // ... copied here (pinned host memory) by the step's last kernel, or nullptr
    unsigned* done_out;          // host frames: [S] pinned words; stream s's workgroup writes done_val behind its (system-fenced) outputs,
    unsigned done_val;           // so that the host can take the frame the moment it is there instead of waiting for the stream to drain
    double* part;                // split form (stream_post_device.h): [S][chunks][21] partial sums, or nullptr (one workgroup per stream)
    unsigned* part_cnt;          // ... and [S] arrival tickets, zero between launches
};
// 64-row chunks of a stack of N rows in the split form: rows 0 .. 63, then 63 per chunk (lane 63 repeats row 0)
static inline int ape_stream_post_chunks(int N) { return N <= 64 ? 1 : 1 + (N - 64 + 62) / 63; }

struct MsgParams {
    const double* est;   // [N,W]
    double* msg;         // [25]
    double body[9];
    int N, W, layout;
};

// Kernel arguments of the Monte-Carlo latency kernel (lstm_mc_small.hip): n_streams windows x n_mc dropout samples, dealt over the
// 8 XCDs -- cluster c serves stream c / cps, sample rows [part * R, part * R + R) of it (part = c % cps).
struct McSmallParams {
    const float* x;                     // windows: stream s at x + s * x_stream_stride, [T][I]
    size_t x_stream_stride;
    float* y;                           // [rows, O] normalised NN targets of the last step
    const float* w0;                    // layer 0: the latency kernel's H/8-member register image (ape_model::wcls[0])
    const float* w[APE_MAX_LAYERS];     // layers >= 1 (index = layer): [member H/8][wave 4][H/16][lane 64][4], the 4x4x1 MFMA's A fragments
    const float* w16[APE_MAX_LAYERS];   // the same layers as 16 x 16 x 4 fragments (16-row clusters): wave = column tile x K half
    const float* bias[APE_MAX_LAYERS];
    const float* w_out;
    const float* b_out;
    const double* xx_m;
    const double* xx_s;
    const double* xx_r;
    char* gx;                           // granules {value, tag}: [cluster 8][parity 2][pair][16 B]
    unsigned gx_cluster_bytes;
    unsigned* seq;                      // [1] launch number of this kernel on this model (upper bits of the tags)
    unsigned* done;                     // [1] departure counter
    unsigned* status;                   // [1] sticky: 1 = a bounded spin gave up
    unsigned* xcc_slots;                // as ClusterParams::xcc_slots: words [192, 448) hold (0x10 | XCC id) of the 8 x 32 workgroups
    const float* masks;                 // injected [L-1, rows, T, H] or nullptr
    int rows;                           // n_streams * n_mc
    int n_mc, n_streams;
    int cps;                            // clusters per stream
    int R;                              // sample rows per cluster (<= 16)
    int T, I, O, x_ring;
    unsigned flags;                     // NORMALIZE_INPUT, DROPOUT_MASKS | DROPOUT_PHILOX, APE_FLAG_ANY_PLACEMENT
    float dropout_p;
    unsigned long long seed;
    unsigned long long* dbg_wg;         // diagnostic builds only
    // ---- the frame's feature builder in the same launch (ape_streams_frame_host, small banks)
    const float* raw_rows;              // [n_streams][raw_width] raw messages, device-visible (pinned host memory will do), or nullptr: the
                                        // windows are complete.  One extra workgroup per stream builds the new row (parse_device.h), hands it
                                        // to the clusters as tagged granules and writes it into the stream's ring for the frames to come
    int raw_width, raw_kind, raw_big_endian;
    int cold;                           // first row after a reset: every step of the window is the new row (estimator.py:96-97)
    float* ring_out;                    // stream 0's first target slot; stream n at + n * ring_stream_stride, copy j at + j * ring_rep_stride
    size_t ring_stream_stride, ring_rep_stride;
    int ring_rep;
    char* xg;                           // feature granules {value, tag}: [n_streams][64] x 8 bytes
};


#ifdef __HIPCC__
// Philox4x32-10 counter-based generator (Salmon et al. 2011) for in-kernel dropout masks.
__device__ __forceinline__ void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                           uint32_t k0, uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        // (one 32 x 32 -> 64 product per multiplier: a single quarter-rate v_mad_u64_u32 instead of v_mul_hi_u32 + v_mul_lo_u32)
        const uint64_t p0 = (uint64_t)0xD2511F53u * (uint64_t)c0, p1 = (uint64_t)0xCD9E8D57u * (uint64_t)c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

#endif

// sets the thread's error string (ape_last_error) and returns `code`: for the entry points outside ape_api.hip
int ape_set_error(int code, const char* msg);

// launchers implemented in the .hip files --------------------------------------------------
// returns hipSuccess or the launch error; `smem_bytes` out for diagnostics
hipError_t ape_launch_lstm_tile16(int H, int L, const LstmParams& p, hipStream_t stream);
size_t ape_lstm_tile16_smem_bytes(int H, int L, int KX, int O, bool dropout);
hipError_t ape_prepare_lstm_tile16(int H, int L, size_t smem_bytes);
hipError_t ape_prepare_lstm_tile16_wide(size_t smem_bytes);
hipError_t ape_prepare_lstm_tile16_upper(int H, int L, size_t smem_bytes);   // layers 1.. of a model on their own: <256,1,wide>, <128,2,wide>
bool ape_cluster_supported(int H, int L, int KX);
bool ape_cluster_layer0_supported(int H, int KX);
hipError_t ape_prepare_lstm_cluster(int H, int L, int KX);
hipError_t ape_launch_lstm_cluster(int H, int L, int KX, int nmt, bool dropout, int clusters, const ClusterParams& p,
                                   hipStream_t stream);
hipError_t ape_launch_lstm_cluster_small(int H, int L, int KX, int nr, int uw, const ClusterParams& p, hipStream_t stream);
hipError_t ape_prepare_lstm_cluster_f16(int H, int L, int KX);
hipError_t ape_launch_lstm_cluster_f16(int H, int L, int KX, int nmt, int clusters, const ClusterParams& p, hipStream_t stream);
bool ape_cluster32_supported(int H, int L, int KX);
hipError_t ape_prepare_lstm_cluster32(int H, int L, int KX);
hipError_t ape_launch_lstm_cluster32(int H, int L, int KX, int clusters, const ClusterParams& p, hipStream_t stream);
// second generation for 16-unit members (lstm_cluster16.hip: the 3 x 128 upper-arm model; first-generation register image wcl)
bool ape_cluster16_supported(int H, int L, int KX);
hipError_t ape_prepare_lstm_cluster16(int H, int L, int KX);
hipError_t ape_launch_lstm_cluster16(int H, int L, int KX, int rows, const ClusterParams& p, hipStream_t stream);
// level-synchronous kernel for short windows of the 3 x 128 model (lstm_level16.hip: 16-window clusters, two workgroups per CU)
bool ape_level16_supported(int H, int L, int KX);
int ape_level16_max_clusters(int n_cus);
size_t ape_level16_gx_bytes(int n_cus);
hipError_t ape_prepare_lstm_level16(int H, int L, int KX);
hipError_t ape_launch_lstm_level16(int H, int L, int KX, int rows, const ClusterParams& p, hipStream_t stream);
bool ape_upper32_supported(int H, int L, int O);
size_t ape_upper32_xfrag_bytes(int rows, int T);
size_t ape_upper32_ypart_bytes(int rows);
size_t ape_lower32_xfrag_bytes(int streams, int T);
size_t ape_lower32_hseq_bytes(int streams, int T);
hipError_t ape_launch_lstm_lower32(const UpperParams& p, const XFragParams& xq, int max_clusters, hipStream_t stream);
hipError_t ape_prepare_lstm_upper32();
// ImuPoseLSTM's 2 x 256 LSTM behind its input layer, one layer per launch: z [B * T][256] row-major -> p0.xfrag (fragment order) ->
// layer 0 (every step to p0.hseq) -> layer 1 reading p1.xfrag == p0.hseq -> head partials -> y [B,O]
hipError_t ape_launch_lstm_split32(const float* z, int B, const UpperParams& p0, const UpperParams& p1, const float* b_out, float* y,
                                   int max_clusters, hipStream_t stream);
hipError_t ape_launch_lstm_upper32(const UpperParams& p, const ExpandParams& q, const float* b_out, float* y, int max_clusters,
                                   hipStream_t stream, hipEvent_t ev_begin = nullptr, hipEvent_t ev_end = nullptr);
bool ape_mc_small_supported(int H, int L, int KX);
size_t ape_mc_small_cluster_bytes(int H, int L);
hipError_t ape_prepare_lstm_mc_small(int H, int L, int KX);
hipError_t ape_launch_lstm_mc_small(int H, int L, int KX, const McSmallParams& p, hipStream_t stream);
bool ape_upper128_supported(int H, int L, int O);
size_t ape_upper128_xfrag_bytes(int rows, int T);
size_t ape_upper128_maskbits_bytes(int rows, int T);
size_t ape_upper128_ypart_bytes(int rows);
size_t ape_upper128_hx_bytes(int clusters);
size_t ape_upper128_flag_words(int clusters);
hipError_t ape_prepare_lstm_upper128();
hipError_t ape_launch_lstm_upper128(const Upper128Params& p, const ExpandParams& q, const float* b_out, float* y, int max_clusters,
                                    hipStream_t stream, hipEvent_t ev_begin = nullptr, hipEvent_t ev_end = nullptr);
bool ape_cluster_f16v2_supported(int H, int L, int KX);
hipError_t ape_prepare_lstm_cluster_f16v2(int H, int L, int KX);
hipError_t ape_launch_lstm_cluster_f16v2(int H, int L, int KX, int clusters, const ClusterParams& p, hipStream_t stream, bool duo = false);
hipError_t ape_launch_parse_rows(const float* rows, int N, int width, int kind, void* out, int out_dtype, int I,
                                 size_t out_stride, int rep, size_t rep_stride, int big_endian, hipStream_t stream);
hipError_t ape_launch_ring_write(const float* xx, int N, int I, float* out, size_t out_stride, int rep,
                                 size_t rep_stride, hipStream_t stream);
hipError_t ape_launch_stream_post(const StreamPostParams& p, hipStream_t stream);
hipError_t ape_launch_mlp_tile16(int H, const MlpParams& p, hipStream_t stream, int n_cus = 0);
hipError_t ape_prepare_mlp_tile16(int H);     // per device: dynamic-LDS limits of the MLP and head-rows kernels (from ape_model_create)
// two-stage weight-stationary pipeline for chip-filling eval batches of the 2-hidden-layer MLP (mlp_pipe.hip)
bool ape_mlp_pipe_supported(int H, int n_hidden, int KX, int O);
size_t ape_mlp_pipe_ring_bytes(int n_cus);
size_t ape_mlp_pipe_ctl_words(int n_cus);
hipError_t ape_prepare_mlp_pipe();
hipError_t ape_launch_mlp_pipe(const MlpParams& q, const float* wa0, const float* wa1, const float* wb2, const float* wbo, float* ring,
                               size_t ring_bytes, unsigned* ctl, int n_cus, hipStream_t stream);
hipError_t ape_launch_head_rows(const float* hseq, int N, int H, int O, const float* w_out, const float* b_out, float* y,
                                hipStream_t stream);
hipError_t ape_launch_fk(const FkParams& p, int preds_dtype, int est_dtype, hipStream_t stream);
hipError_t ape_launch_msg_reduce(const MsgParams& p, hipStream_t stream);
