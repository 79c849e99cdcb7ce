/*
 * ape_hip.h -- C ABI of libape_hip.so: the MI355X (gfx950) implementation of the per-frame
 * arm-pose inference path of wear_mocap_ape.
 *
 * The reference (pure Python) has no FFI; its boundary for this path is the Python
 * `Estimator` template-method contract.  The entry points below are
 * exactly what a reference-side ctypes binding for that path would call; each one names the
 * reference code it replaces (paths relative to /root/reference/src/wear_mocap_ape).  The
 * binding a maintainer would add is shown in INTEGRATION.md; the host-side mirror of the
 * reference classes that uses it lives in arm-pose-estimation_amd/wear_mocap_ape_amd/.
 *
 * Conventions
 *   - plain C: opaque handle, raw pointers, sizes; no torch / C++ types.
 *   - every `*_dev` pointer is DEVICE memory on the model's GPU (e.g. tensor.data_ptr()),
 *     row-major, contiguous.  `stream` is a hipStream_t passed as void* (NULL = default
 *     stream).  Calls enqueue work on `stream` and return without synchronising; they
 *     perform no allocation once `ape_model_reserve` covers the batch (graph-capture safe).
 *   - return value 0 = success; anything else is an APE_ERR_* code and `ape_last_error()`
 *     (thread-local) describes it.  The Python mirror raises `UserWarning` for a non-zero
 *     status, the reference's exception convention (nn_models.py:385-400, transformations.py:98-116).
 *   - ONE model handle serialises on ONE stream at a time: the handle owns the exchange buffers, flag words, arrival
 *     tickets and workspaces its kernels use, so two calls on the same handle (ape_lstm_forward, ape_infer,
 *     ape_streams_step of any bank built on it) must not run concurrently -- enqueue them on the same stream, or
 *     order the streams with events.  Use one handle per concurrently running stream/thread (reference: one
 *     Estimator, with its own model, per consumer thread, estimator.py:139-143).  Different handles are independent.
 *   - a launch of the weight-stationary kernels that cannot make progress (its workgroups never all resident, e.g. on
 *     a GPU shared with other long-running kernels) gives up after a bounded wait, sets a sticky status word and
 *     leaves its outputs unwritten; every later launch on that handle then leaves at once, also without writing.
 *     ape_model_check() reports (and clears) that state; ape_model_recover() does the same and then RE-ISSUES every
 *     call made on the handle since its last successful check on the kernels that need no co-residency (the batch-tile
 *     LSTM / MLP kernels), on the streams the calls named, so that no frame is lost: call one of them wherever results
 *     are consumed on the host (the Python mirror calls ape_model_recover whenever it copies results to host memory).
 *   - quaternions are [w,x,y,z]; all joint/column indices are fixed by the layouts below.
 */
#ifndef APE_HIP_H
#define APE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define APE_ABI_VERSION 7

/* ---- status codes ---------------------------------------------------------------------- */
enum {
    APE_OK = 0,
    APE_ERR_INVALID_ARG = 1,   /* NULL pointer, non-positive size, unknown enum value           */
    APE_ERR_UNSUPPORTED = 2,   /* dims outside what the gfx950 kernels are built for            */
    APE_ERR_NOT_READY = 3,     /* weights / norm stats not loaded yet                           */
    APE_ERR_HIP = 4,           /* a HIP runtime call failed (message carries hipGetErrorString) */
    APE_ERR_NO_DEVICE = 5,     /* no usable gfx950 device: there is NO CPU fallback             */
    APE_ERR_CAPACITY = 6       /* batch larger than ape_model_reserve()d workspace during capture */
};

/* ---- NN-target layouts: utility/names.py:4-29 (NNS_TARGETS) -------------------------------
 * est row layouts: estimate/estimate_joints.py:48-71 / :74-92 / :20-45
 *   APE_LAYOUT_ORI_CAL_LARM_UARM_HIPS      O=14 -> est[21] = hand(0:3) larm_orig(3:6) uarm_orig(6:9)
 *                                                           larm_q(9:13) uarm_q(13:17) hips_q(17:21)
 *   APE_LAYOUT_ORI_CAL_LARM_UARM           O=12 -> est[14] = hand(0:3) larm_orig(3:6) larm_q(6:10) uarm_q(10:14)
 *   APE_LAYOUT_ORI_POS_CAL_LARM_UARM_HIPS  O=20 -> est[21] (same columns as the first)
 * message layout (all three): estimate/compose_msg.py:72-78
 *   msg[25] = hand_rot(0:4, == larm_rot) hand_orig(4:7) larm_rot(7:11) larm_orig(11:14)
 *             uarm_rot(14:18) uarm_orig(18:21) hips_rot(21:25)
 */
enum {
    APE_LAYOUT_NONE = -1,                    /* regressor only: ape_fk / ape_msg_reduce / ape_infer refuse */
    APE_LAYOUT_ORI_CAL_LARM_UARM_HIPS = 0,
    APE_LAYOUT_ORI_CAL_LARM_UARM = 1,
    APE_LAYOUT_ORI_POS_CAL_LARM_UARM_HIPS = 2
};

enum { APE_F32 = 0, APE_F64 = 1 };   /* element type selector for preds / est buffers */

/* ---- flags for ape_lstm_forward / ape_infer ---------------------------------------------- */
#define APE_FLAG_NORMALIZE_INPUT 0x1u /* x is raw features: z-score in f64, cast f32 (estimator.py:103-104,
                                         watch_phone_pocket_nn.py:100); else x is already normalised  */
#define APE_FLAG_ALL_STEPS       0x2u /* y is [B,T,O] like DropoutLSTM.forward (nn_models.py:188-189);
                                         else only the last step [B,O] (watch_phone_pocket_nn.py:111)   */
#define APE_FLAG_DROPOUT_MASKS   0x4u /* inter-layer dropout with caller-supplied masks (train-mode LSTM
                                         after monte_carlo_predictions, nn_models.py:204)               */
#define APE_FLAG_DROPOUT_PHILOX  0x8u /* inter-layer dropout with an in-kernel counter-based generator  */

/* LSTM kernel selection (ape_model_set_kernel).  AUTO = the weight-stationary cluster kernel where it is
 * built (H=256/L=2/I<=32 and H=128/L=3/32<I<=64; dropout up to 32 windows per cluster), else the batch-tile kernel. */
enum { APE_KERNEL_AUTO = 0, APE_KERNEL_TILE16 = 1, APE_KERNEL_CLUSTER = 2,
       APE_KERNEL_CLUSTER_GEN1 = 3 /* the cluster kernels, first generation only: A/B runs against lstm_cluster32.hip */,
       APE_KERNEL_AUTO_GEN1 = 4    /* AUTO's dispatch without the second-generation kernels (lstm_cluster32.hip, and
                                      lstm_upper32.hip in a Monte-Carlo stream bank): A/B runs, tests */ };

/* Storage precision of W, x and h inside the LSTM (ape_model_set_precision).  F32 (default): exact float32
 * MFMA.  F16: binary16 weights / inputs / hidden state with float32 accumulate, cell state and head
 * (BASELINE.json configs[4]); last-step output without dropout; parity to a stated tolerance only.
 * F16_GEN1: the same arithmetic on the first-generation fp16 kernel for every batch size (A/B runs, tests); F16
 * serves batches above 256 rows of the 2 x 256 models with the row-set-pipelined kernel. */
enum { APE_PRECISION_F32 = 0, APE_PRECISION_F16 = 1, APE_PRECISION_F16_GEN1 = 2 };

#define APE_FLAG_PACKED_MSG      0x20u /* ape_streams_step only: message and tail of a stream packed in one row  */
#define APE_FLAG_BROADCAST_X     0x10u /* x_dev is ONE window [1,T,I] shared by all B rows: the x.repeat((n,1,1)) of
                                         monte_carlo_predictions (nn_models.py:206) without materialising it    */

/* Exchange-form selectors of the weight-stationary kernels, for A/B runs and tests (ape_lstm_forward, ape_infer, ape_streams_step).  They
 * change HOW the workgroups of a cluster hand their slices over, never the arithmetic: results are bit-equal (ALT_FORM: equal up to
 * float32 summation order).  Every other undeclared bit is refused.
 * DEFAULT hand-over of every flag-based kernel (round 5): payload by write-through (`sc1`) stores, every storing wave waits for its
 * stores, then its flag (an agent-scope store); every load of handed-off bytes is an `sc1` load -- the form the MI355X guide lists as
 * valid wherever the workgroups run. */
#define APE_FLAG_ANY_PLACEMENT   0x08000000u /* the two tagged-granule latency kernels (lstm_cluster_small / lstm_mc_small): write-through
                                               granule stores also where the members share an XCD.  A no-op for the flag-based kernels,
                                               whose default it names since round 5 */
#define APE_FLAG_IN_XCD_PLAIN    0x00400000u /* OPT-IN, A/B runs and tests only: payload by plain (write-back) stores where all members of a
                                               cluster were verified to share an XCD (the default until round 4; 1-2 % faster in float32,
                                               10 % in fp16).  OUTSIDE the guide's table of valid forms: a plain store behind
                                               `s_waitcnt vmcnt(0)` is no agent-scope release.  Same bits when it holds */
#define APE_FLAG_NO_XCD_CLASSES  0x02000000u /* first-generation cluster kernel: clusters by global arrival ticket instead of within
                                               block-index classes (one XCD each) */
#define APE_FLAG_ALT_FORM        0x01000000u /* the alternative decomposition where a kernel has two: the latency kernel's H/16-member form,
                                               lstm_cluster16's one-workgroup-per-CU form, the fp16 kernel's 16-unit-member form (two
                                               workgroups per CU; measured slower), ImuPoseLSTM's one-tile clusters on the
                                               blocking exchange instead of the gather under the input span (same bits) */

typedef struct ape_model ape_model_t;

/* DropoutLSTM(input_size, hidden_layer_size, hidden_layer_count, output_size) -- nn_models.py:160-178,
 * constructed as load_deployed_model_from_hash does (nn_models.py:402-408). */
/* regressor architectures the reference loader dispatches (nn_models.py:393-400) */
enum {
    APE_MODEL_LSTM = 0,   /* DropoutLSTM  nn_models.py:160-207 */
    APE_MODEL_FF = 1,     /* DropoutFF    nn_models.py:313-370 : Linear, n x Linear (leaky_relu), dropout, Linear */
    APE_MODEL_IMUPOSE = 2 /* ImuPoseLSTM  nn_models.py:210-249 : Linear(I,256)+ReLU, fixed 2 x 256 LSTM, Linear(256,O);
                             hidden_size must be 256 and num_layers 2 (the reference ignores its ctor arguments) */
};

typedef struct ape_dims {
    int32_t input_size;    /* I: 20 / 22 / 38 (<= 64)                      */
    int32_t hidden_size;   /* H: 128 or 256                                */
    int32_t num_layers;    /* L: 1..3                                      */
    int32_t output_size;   /* O: 12 / 14 / 20 (<= 32)                      */
    int32_t target_layout; /* APE_LAYOUT_*; O must match it (unless NONE)  */
    int32_t device;        /* HIP device ordinal                           */
    int32_t model_kind;    /* APE_MODEL_*; for APE_MODEL_FF num_layers is hidden_layer_count (0..7) */
} ape_dims_t;

/* library / device --------------------------------------------------------------------------- */
int ape_abi_version(void);
const char* ape_last_error(void);
/* number of visible HIP devices whose arch is gfx950 (0 on a CPU-only host; never fails) */
int ape_device_count(void);

/* lifetime ------------------------------------------------------------------------------------ */
/* replaces: nn_models.DropoutLSTM.__init__ (nn_models.py:161-178) + Estimator.__init__ defaults
 * (estimator.py:57-68: default body measurements are installed) */
int ape_model_create(const ape_dims_t* dims, ape_model_t** out_model);
int ape_model_destroy(ape_model_t* model);
/* pre-allocate the [max_batch, O] intermediate so that later calls do not allocate */
int ape_model_reserve(ape_model_t* model, int32_t max_batch);

/* replaces: nn_model.load_state_dict(model_state) (nn_models.py:410-411).
 * `blob` = float32 tensors concatenated in state_dict order:
 *   for k in 0..L-1: lstm.weight_ih_l{k} [4H, I or H], lstm.weight_hh_l{k} [4H,H],
 *                    lstm.bias_ih_l{k} [4H], lstm.bias_hh_l{k} [4H];
 *   then output_layer.weight [O,H], output_layer.bias [O].
 * APE_MODEL_FF: _input_layer.weight [H,I], .bias [H]; _hidden_layers.{k}.weight [H,H], .bias [H] for k in
 *   0..hidden_layer_count-1; _output_layer.weight [O,H], .bias [O]   (state_dict order of DropoutFF).
 * APE_MODEL_IMUPOSE: input_layer.weight [256,I], .bias [256]; then the APE_MODEL_LSTM tensors with a 256-wide
 *   layer-0 input (state_dict order of ImuPoseLSTM).
 * `blob` may be host or device memory (e.g. the buffer an RCCL broadcast just filled);
 * `n_floats` must equal ape_weight_blob_floats(dims).  Synchronous; init-time only. */
int ape_model_load_weights(ape_model_t* model, const float* blob, size_t n_floats);
size_t ape_weight_blob_floats(const ape_dims_t* dims);

/* replaces: Estimator.__init__ stats load / Estimator.set_norm_stats (estimator.py:35-42,72-77).
 * Host pointers, float64: xx_m, xx_s [I]; yy_m, yy_s [O]. */
int ape_model_set_norm_stats(ape_model_t* model, const double* xx_m, const double* xx_s,
                             const double* yy_m, const double* yy_s);
/* replaces: Estimator._body_measurements (estimator.py:57-68): [larm_vec(3), uarm_vec(3), uarm_orig_rh(3)] */
int ape_model_set_body(ape_model_t* model, const double body9[9]);

/* hot path ------------------------------------------------------------------------------------ */
/* replaces: DropoutLSTM.forward / monte_carlo_predictions (nn_models.py:180-207) as called from
 * make_prediction_from_row_hist (watch_phone_pocket_nn.py:98-112, watch_only.py:84-97,
 * watch_phone_uarm_nn.py:107-121).  h0 = c0 = 0 for every window (hs=None).
 *   x_dev      f32 [B,T,I]
 *   masks_dev  f32 [L-1,B,T,H] holding 0 or 1/(1-p)   (APE_FLAG_DROPOUT_MASKS; else NULL)
 *   dropout_p, seed                                   (APE_FLAG_DROPOUT_PHILOX)
 *   y_dev      f32 [B,O] or [B,T,O] (APE_FLAG_ALL_STEPS): normalised NN targets
 * APE_MODEL_FF (DropoutFF.forward / monte_carlo_predictions, nn_models.py:340-370): the MLP is applied to the
 *   last step of every window (or to all B*T rows with APE_FLAG_ALL_STEPS); masks_dev is f32 [rows,H], the
 *   dropout in front of the output layer.
 * APE_MODEL_IMUPOSE (ImuPoseLSTM.forward, nn_models.py:236-244): as APE_MODEL_LSTM, no dropout modes (its
 *   monte_carlo_predictions is the plain forward, :246-251). */
int ape_lstm_forward(ape_model_t* model, const float* x_dev, int32_t B, int32_t T, uint32_t flags,
                     const float* masks_dev, float dropout_p, uint64_t seed,
                     float* y_dev, void* stream);
/* the same with a caller-given initial state: DropoutLSTM.forward(x, hs=(h0, c0)) (nn_models.py:180-189 hands hs to
 * nn.LSTM).  h0_dev, c0_dev: f32 [L,B,H] (both or neither; NULL, NULL = ape_lstm_forward).  Not for APE_MODEL_FF
 * and not with the fp16 variant. */
int ape_lstm_forward_hs(ape_model_t* model, const float* x_dev, int32_t B, int32_t T, uint32_t flags,
                        const float* masks_dev, float dropout_p, uint64_t seed,
                        const float* h0_dev, const float* c0_dev, float* y_dev, void* stream);

/* replaces: estimate_joints.arm_pose_from_nn_targets (estimate_joints.py:16-17) and, with
 * `denormalize` != 0, the `pred * yy_s + yy_m` of estimator.py:108-109 in front of it.
 *   preds_dev  [N,O] of preds_dtype;  est_dev [N,W] of est_dtype (W = 21 or 14).
 * All arithmetic is float64 on the device whatever the storage types. */
int ape_fk(ape_model_t* model, const void* preds_dev, int32_t preds_dtype, int32_t N,
           int32_t denormalize, void* est_dev, int32_t est_dtype, void* stream);

/* replaces: compose_msg.msg_from_nn_targets_est (compose_msg.py:13-14): N est rows -> msg[25].
 * N > 1: sign-aligned quaternion means (transformations.py:32-51) and origins recomputed from
 * them; N == 1: row 0 copied into the message layout.  est_dev f64 [N,W]; msg_dev f64 [25]. */
int ape_msg_reduce(ape_model_t* model, const double* est_dev, int32_t N, double* msg_dev, void* stream);

/* feature builder, batched.  replaces: parse_row_to_xx of WatchPhonePocketNN
 * (watch_phone_pocket_nn.py:41-96), WatchOnlyNN (watch_only.py:46-82; also with the 55-float watch+phone
 * message, watch_only.py:29-32) and WatchPhoneUarmNN (watch_phone_uarm_nn.py:43-105).
 *   rows_dev f32 [N,55] (or [N,28] for APE_PARSE_WATCH_ONLY): raw messages, data_types/messaging.py layouts
 *   xx_dev   [N,22|20|20|38] of xx_dtype (the reference returns float32, float64 for the upper-arm variant)
 * float64 arithmetic on the device.  Needs no model: runs on the current HIP device. */
enum {
    APE_PARSE_WATCH_PHONE_POCKET = 0,   /* 55 -> 22 */
    APE_PARSE_WATCH_ONLY = 1,           /* 28 -> 20 */
    APE_PARSE_WATCH_ONLY_PHONE_MSG = 2, /* 55 -> 20 */
    APE_PARSE_WATCH_PHONE_UARM = 3,     /* 55 -> 38 */
    /* OR-ed into a kind: the rows are UDP payloads as received -- big-endian float32
     * (stream_listener/imu.py:53,68-69 unpacks them with '>f'); default is native float32 */
    APE_PARSE_BIG_ENDIAN = 0x100
};
int ape_parse_rows(int32_t kind, const float* rows_dev, int32_t N, void* xx_dev, int32_t xx_dtype, void* stream);

/* the whole batched path in one call: x -> [normalise] -> LSTM -> last step ->
 * de-normalise -> FK.  y_dev (f32 [B,O], normalised NN targets) may be NULL. */
/* stream bank: the per-frame step of S independent wearable streams, state resident on the device (SURVEY.md 8a-1,
 * 8a-15, 8f-2).  replaces, for all streams at once, Estimator.add_xx_to_row_hist_and_make_prediction
 * (estimator.py:93-120: window of the last seq_len feature rows, padded with the newest row on a cold start :96-97;
 * z-score; model; de-normalise; smoothing stack of the last `smooth` predictions, padded the same way :112-118) and
 * Estimator.msg_from_pred (:122-137) with one Monte-Carlo sample per stream (deterministic weights).
 *   ape_streams_push_rows      rows_dev f32 [S,55|28] raw messages of `kind` (ape_parse_rows kinds, may carry
 *                              APE_PARSE_BIG_ENDIAN) -> features -> next slot of every stream's window ring
 *   ape_streams_push_features  xx_dev f32 [S,I]: the same for callers that build features themselves
 *   ape_streams_step           one prediction per stream from the current windows:
 *                              msg_dev  [S,25] of out_dtype, layout of compose_msg.py:72-78
 *                              tail_dev [S,smooth,6] of out_dtype or NULL: hand and elbow xyz of every smoothing row
 *                              (what msg_from_pred appends to the message when add_mc_samples is set and smooth > 1)
 *                              flags: APE_FLAG_NORMALIZE_INPUT and / or APE_FLAG_PACKED_MSG.  PACKED_MSG:
 *                              msg_dev is [S, 25+6N] of out_dtype (N = smooth*n_mc stacked rows), every row the
 *                              message followed by its tail = the list Estimator.msg_from_pred returns
 *                              (estimator.py:131-137); as APE_F32 it is byte for byte the payload the reference sends
 *                              per estimator (pose_est_udp.py:47 struct.pack('f'*len(msg))); tail_dev must be NULL
 *   ape_streams_reset          cold start: the next row fills the whole window, the next prediction the whole stack
 *   ape_streams_set_mc         Monte-Carlo dropout per stream, as every reference estimator runs it
 *                              (monte_carlo_samples, watch_phone_pocket_nn.py:105-110 -> nn_models.py:191-207): each
 *                              frame runs every stream's window n_mc times with independent inter-layer dropout
 *                              masks (in-kernel Philox, keyed by `seed` + a per-step counter), the smoothing stack
 *                              holds smooth x n_mc rows per stream in the reference's order (estimator.py:112-118:
 *                              oldest prediction first, its n_mc samples in order), tail_dev becomes
 *                              [S, smooth*n_mc, 6].  dropout_p = 0 gives n_mc identical samples.  Call it before the
 *                              first row is pushed or right after ape_streams_reset (it re-allocates the rings and
 *                              implies a reset); smooth*n_mc <= 4096.
 * One bank = one model handle = one HIP stream at a time.  smooth <= 64. */
typedef struct ape_streams ape_streams_t;
int ape_streams_create(ape_model_t* model, int32_t n_streams, int32_t seq_len, int32_t smooth, ape_streams_t** out_bank);
int ape_streams_destroy(ape_streams_t* bank);
int ape_streams_reset(ape_streams_t* bank);
int ape_streams_set_mc(ape_streams_t* bank, int32_t n_mc, float dropout_p, uint64_t seed);
int ape_streams_push_rows(ape_streams_t* bank, int32_t kind, const float* rows_dev, void* stream);
int ape_streams_push_features(ape_streams_t* bank, const float* xx_dev, void* stream);
int ape_streams_step(ape_streams_t* bank, uint32_t flags, void* msg_dev, void* tail_dev, int32_t out_dtype, void* stream);
/* ONE iteration of Estimator.processing_loop (estimator.py:174-177: parse_row_to_xx -> add_xx_to_row_hist_and_make_prediction
 * -> msg_from_pred) for every stream of the bank, with HOST buffers (ABI 6): what the drop-in Estimator classes call per frame.
 *   rows_host  f32 [S,55|28] raw messages of `kind` (may carry APE_PARSE_BIG_ENDIAN), ordinary host memory
 *   out_host   [S, 25+6N] of out_dtype (N = smooth*n_mc): message + tail of every stream, ordinary host memory
 *   flags      APE_FLAG_NORMALIZE_INPUT or 0
 * = ape_streams_push_rows + ape_streams_step(PACKED_MSG) + the copies either side, BLOCKING: the rows travel through pinned
 * staging the kernels read and write directly (no copy command on the stream), the call returns when out_host is filled.
 * The health of the frame's launches is part of the frame: an aborted weight-stationary launch is re-issued as by
 * ape_model_recover before the call returns, and a clean frame clears the handle's journal. */
int ape_streams_frame_host(ape_streams_t* bank, int32_t kind, const float* rows_host, uint32_t flags, void* out_host,
                           int32_t out_dtype, void* stream);
/* where a host frame's time goes (ABI 7; bench.py `batch1.estimator_loop`): ape_streams_frame_host keeps, for the last 4096 frames, the
 * host time spent (us) in [0] the rows' copy into pinned staging + the frame's launch calls, [1] the wait from the last launch call's
 * return to all completion words seen (or the stream synchronised), [2] the copy of the pinned output into out_host; and counts the frames
 * whose completion words were NOT seen within the poll budget and fell through to hipStreamSynchronize.  `trace_us` (may be NULL):
 * [n][3] floats, oldest frame first, n = min(frames since the last reset, capacity_frames, 4096) -> *n_out.  reset != 0 clears. */
typedef struct ape_frame_stats {
    uint64_t frames;           /* ape_streams_frame_host calls since the last reset        */
    uint64_t fallback_syncs;   /* ... of them, frames that ended in hipStreamSynchronize   */
    uint64_t recovered;        /* ... frames whose cooperative launch gave up and was re-issued */
} ape_frame_stats_t;
int ape_streams_frame_stats(ape_streams_t* bank, ape_frame_stats_t* out, float* trace_us, int32_t capacity_frames, int32_t* n_out,
                            int32_t reset);
/* measurement aid (bench.py `stream_bank_T6.*.roofline`): with profiling on, every launch of the step's dominant kernel (the
 * regressor: ape_lstm_upper32 in a Monte-Carlo bank, else the LSTM launch) is bracketed by a pair of HIP events on the
 * step's own stream; ape_streams_profile_read synchronises, returns the summed duration and the number of launches since
 * the last read, and re-arms.  At most 256 launches are recorded between two reads; off by default (two event records
 * per launch otherwise). */
int ape_streams_profile(ape_streams_t* bank, int32_t enable);
int ape_streams_profile_read(ape_streams_t* bank, double* kernel_ms_sum, int32_t* launches);

int ape_infer(ape_model_t* model, const float* x_dev, int32_t B, int32_t T, uint32_t flags,
              float* y_dev, void* est_dev, int32_t est_dtype, void* stream);

/* kernel selection for A/B runs and tests; no effect on results beyond float32 summation order */
int ape_model_set_kernel(ape_model_t* model, int32_t choice);
int ape_model_set_precision(ape_model_t* model, int32_t precision);
/* BLOCKING health check (hipDeviceSynchronize, i.e. every stream of the device): non-zero if a cluster-kernel launch
 * since the last check gave up waiting for a peer workgroup (its bounded spins expired) -- the outputs of that launch
 * and of every later one on this handle are invalid.  A failing check also resets the handle: the next launch works. */
int ape_model_check(ape_model_t* model);
/* ape_model_check that survives an abort.  The handle keeps a journal of the compute calls made on it since the last
 * successful check / recover (ape_lstm_forward[_hs], ape_fk, ape_msg_reduce, ape_infer, ape_streams_step; up to 64).  When
 * the blocking check finds an aborted weight-stationary launch, the handle is reset as by ape_model_check and every
 * journaled call is issued again, in order, on its own stream, with the cooperative kernels switched off (batch-tile LSTM
 * kernel, tile MLP kernel: no workgroup waits for another; same arithmetic up to float32 summation order, so results
 * agree with the aborted kernels' to ~1e-6 -- an fp16-precision model is re-run in exact float32), then the device is
 * synchronised: 0 = nothing was aborted, or everything was re-issued and the outputs are valid now.  The CALLER'S PART:
 * the device buffers those calls read must still hold the same data (recover before re-using them; a Python mirror that
 * copies outputs to the host right behind a call satisfies this by construction).  A stream-bank step can be re-issued
 * only while it is the bank's newest step and no row was pushed behind it; otherwise, or when more than 64 calls are
 * pending, the function returns APE_ERR_HIP like ape_model_check and counts the calls as lost.  Never a CPU path: the
 * re-issue is a fresh launch of HIP kernels in this process. */
int ape_model_recover(ape_model_t* model);
typedef struct ape_model_stats {
    uint64_t aborted_checks;      /* checks / recovers that found an aborted launch                       */
    uint64_t reissued_calls;      /* calls ape_model_recover issued again on the non-cooperative kernels  */
    uint64_t lost_calls;          /* calls pending at an abort that could not be re-issued                */
} ape_model_stats_t;
int ape_model_stats(const ape_model_t* model, ape_model_stats_t* out);

/* introspection for benchmarks: name of the dominant kernel for (B,T) and its algorithmic
 * FLOP per window (sum_layers 2*4H*(in_l+H) per step, + 2*O*H head once). */
const char* ape_lstm_kernel_name(const ape_model_t* model, int32_t B, int32_t T);
/* the LSTM kernel the newest call on this handle launched last ("ape_lstm_mc_small", "ape_lstm_cluster32", "ape_lstm_tile16", ...;
 * "" before the first call): tests and benchmarks assert the route they mean to measure */
const char* ape_model_last_kernel(const ape_model_t* model);
double ape_flops_per_window(const ape_dims_t* dims, int32_t T);

/* ---- ensemble Kalman estimator (tail; ABI 4) ----------------------------------------
 * Replaces KalmanSmartwatchModel (reference estimate/kalman_models.py:139-220: ProcessModelWindow :8-50,
 * ObservationNoise :53-80, SensorModelWindow :83-136) behind WatchPhonePocketKalman (watch_phone_pocket_kalman.py:12-169).
 * PARITY UNPINNED: the reference module imports bayesian_torch (absent) and its checkpoint is absent; the checker is
 * oracle/kalman_oracle.py, a restatement of the cited lines and of the published LinearFlipout algorithm.
 *
 * Rows are (stream s, ensemble member e), batch-major.  One forward = one draw of the flipout weight perturbations,
 * shared by all rows of the call (LinearFlipout draws eps once per call), and per-element +-1 signs; the Kalman update
 * (means, observation noise, 14 x 14 innovation, inverse, gain: kalman_models.py:181-208) is per stream, i.e. the
 * reference's batch size 1 (watch_phone_pocket_kalman.py:135) for every stream.
 *
 * Weight blob (float32, ape_kalman_weight_floats values), layers in this order with the reference's names:
 *   process_model.bayes1, process_model.bayes3 (flipout), process_model.bayes_m2 (linear),
 *   sensor_model.fc2 (linear), sensor_model.fc3, .fc5, .fc6 (flipout), observation_noise.fc1, .fc2 (linear);
 *   a flipout layer contributes mu_weight [N,K], rho_weight [N,K], mu_bias [N], rho_bias [N]; a linear one weight [N,K], bias [N].
 * noise_dev: NULL (device-side Philox draws from `seed`) or ape_kalman_noise_floats(S) floats of injected draws, per flipout
 *   layer in blob order: eps_weight [N,K], eps_bias [N] (standard normal), sign_in [S*E,K], sign_out [S*E,N] (+-1).
 * win_size must be even (rows are read 16 bytes at a time). */
typedef struct ape_kalman ape_kalman_t;
typedef struct {
    int32_t num_ensemble;   /* E: ensemble members (watch_phone_pocket_kalman.py:16; 2..128) */
    int32_t win_size;       /* W: window of previous states / raw observations (:17) */
    int32_t device;
} ape_kalman_dims_t;
int ape_kalman_create(const ape_kalman_dims_t* dims, ape_kalman_t** out_model);
int ape_kalman_destroy(ape_kalman_t* model);
size_t ape_kalman_weight_floats(const ape_kalman_t* model);
size_t ape_kalman_noise_floats(const ape_kalman_t* model, int32_t S);
int ape_kalman_load_weights(ape_kalman_t* model, const float* blob_host, size_t n_floats);
/* KalmanSmartwatchModel.forward (kalman_models.py:175-220): raw_obs [S,W,22], state_prev [S,E,W,14] ->
 * state_corrected [S,E,14], m_state_corrected [S,14], m_state_pred [S,14], z [S,14], ensemble_z [S,E,14] (all device, f32) */
int ape_kalman_forward(ape_kalman_t* model, const float* raw_obs_dev, const float* state_prev_dev, int32_t S, uint64_t seed,
                       const float* noise_dev, float* state_corrected_dev, float* m_state_corrected_dev,
                       float* m_state_pred_dev, float* z_dev, float* ensemble_z_dev, void* stream);
/* KalmanSmartwatchModel.format_state (kalman_models.py:164-173): state [S,14] -> [S,E,14] = state + N(0, 0.1 I);
 * noise_dev NULL or [S,E,14] standard-normal draws */
int ape_kalman_format_state(ape_kalman_t* model, const float* state_dev, int32_t S, uint64_t seed, const float* noise_dev,
                            float* out_dev, void* stream);
/* BLOCKING: non-zero if a forward since the last check met an exactly singular innovation matrix (torch.linalg.inv raises) */
int ape_kalman_check(ape_kalman_t* model);

#ifdef __cplusplus
}
#endif
#endif /* APE_HIP_H */
