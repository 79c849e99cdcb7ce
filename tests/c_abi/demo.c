/* A caller of libape_hip.so that is neither Python nor PyTorch: plain C against include/ape_hip.h and the HIP runtime.
 * Builds the pocket regressor from a seeded weight blob, pushes seeded windows through ape_infer (z-score + LSTM +
 * head + de-normalise + FK), through the stream bank and through the one-call host frame (ABI 6, with its ABI-7 statistics), and writes the results to a file that
 * tests/test_hip_parity.py::test_c_caller compares with the Python binding on the same numbers.
 *
 *   gcc -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude tests/c_abi/demo.c \
 *       -Larm-pose-estimation_amd/lib -lape_hip -L/opt/rocm/lib -lamdhip64 -o demo
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "ape_hip.h"

#define CHECK_APE(call)                                                                         \
    do {                                                                                        \
        int rc_ = (call);                                                                       \
        if (rc_ != APE_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, ape_last_error()); return 2; } \
    } while (0)
#define CHECK_HIP(call)                                                                         \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #call, hipGetErrorString(e_)); return 3; } \
    } while (0)

/* 32-bit LCG (Numerical Recipes constants) -> float in [-1, 1): trivially reproducible in Python */
static uint32_t lcg_state;
static float lcg_uniform(void) {
    lcg_state = lcg_state * 1664525u + 1013904223u;
    return (float)(lcg_state >> 8) * (2.0f / 16777216.0f) - 1.0f;
}

int main(int argc, char** argv) {
    const char* out_path = argc > 1 ? argv[1] : "c_abi_demo.bin";
    enum { I = 22, H = 256, L = 2, O = 14, B = 37, T = 6, S = 5, SMOOTH = 3 };
    if (ape_abi_version() != APE_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 1; }
    if (ape_device_count() < 1) { fprintf(stderr, "no gfx950 device\n"); return 1; }

    ape_dims_t dims = {I, H, L, O, APE_LAYOUT_ORI_CAL_LARM_UARM_HIPS, 0, APE_MODEL_LSTM};
    ape_model_t* model = NULL;
    CHECK_APE(ape_model_create(&dims, &model));

    const size_t n_w = ape_weight_blob_floats(&dims);
    float* blob = (float*)malloc(n_w * sizeof(float));
    lcg_state = 12345u;
    for (size_t i = 0; i < n_w; ++i) blob[i] = 0.0625f * lcg_uniform();            /* +-1/sqrt(H) */
    CHECK_APE(ape_model_load_weights(model, blob, n_w));

    double xx_m[I], xx_s[I], yy_m[O], yy_s[O];
    for (int i = 0; i < I; ++i) { xx_m[i] = 0.1 * i - 1.0; xx_s[i] = 0.5 + 0.05 * i; }
    for (int i = 0; i < O; ++i) { yy_m[i] = 0.02 * i; yy_s[i] = 0.3 + 0.01 * i; }
    CHECK_APE(ape_model_set_norm_stats(model, xx_m, xx_s, yy_m, yy_s));
    const double body[9] = {-0.22, 0, 0, -0.26, 0, 0, -0.1704612, 0.4309841, -0.00670862};
    CHECK_APE(ape_model_set_body(model, body));

    /* windows: raw features around the statistics */
    float* x = (float*)malloc((size_t)B * T * I * sizeof(float));
    lcg_state = 777u;
    for (int b = 0; b < B; ++b)
        for (int t = 0; t < T; ++t)
            for (int i = 0; i < I; ++i) x[((size_t)b * T + t) * I + i] = (float)(xx_m[i] + xx_s[i] * lcg_uniform());
    float *x_dev, *y_dev;
    double* est_dev;
    CHECK_HIP(hipMalloc((void**)&x_dev, (size_t)B * T * I * sizeof(float)));
    CHECK_HIP(hipMalloc((void**)&y_dev, (size_t)B * O * sizeof(float)));
    CHECK_HIP(hipMalloc((void**)&est_dev, (size_t)B * 21 * sizeof(double)));
    CHECK_HIP(hipMemcpy(x_dev, x, (size_t)B * T * I * sizeof(float), hipMemcpyHostToDevice));
    hipStream_t stream;
    CHECK_HIP(hipStreamCreate(&stream));
    CHECK_APE(ape_infer(model, x_dev, B, T, APE_FLAG_NORMALIZE_INPUT, y_dev, est_dev, APE_F64, stream));
    CHECK_HIP(hipStreamSynchronize(stream));
    CHECK_APE(ape_model_check(model));
    float* y = (float*)malloc((size_t)B * O * sizeof(float));
    double* est = (double*)malloc((size_t)B * 21 * sizeof(double));
    CHECK_HIP(hipMemcpy(y, y_dev, (size_t)B * O * sizeof(float), hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(est, est_dev, (size_t)B * 21 * sizeof(double), hipMemcpyDeviceToHost));

    /* stream bank: S streams, feature rows pushed frame by frame (the first T frames of the first S windows) */
    ape_streams_t* bank = NULL;
    CHECK_APE(ape_streams_create(model, S, T, SMOOTH, &bank));
    float* xx_dev;
    double* msg_dev;
    CHECK_HIP(hipMalloc((void**)&xx_dev, (size_t)S * I * sizeof(float)));
    CHECK_HIP(hipMalloc((void**)&msg_dev, (size_t)S * 25 * sizeof(double)));
    float frame[S * I];
    for (int t = 0; t < T; ++t) {
        for (int s = 0; s < S; ++s)
            for (int i = 0; i < I; ++i) frame[s * I + i] = x[((size_t)s * T + t) * I + i];
        CHECK_HIP(hipMemcpyAsync(xx_dev, frame, sizeof(frame), hipMemcpyHostToDevice, stream));
        CHECK_HIP(hipStreamSynchronize(stream));
        CHECK_APE(ape_streams_push_features(bank, xx_dev, stream));
        CHECK_APE(ape_streams_step(bank, APE_FLAG_NORMALIZE_INPUT, msg_dev, NULL, APE_F64, stream));
    }
    CHECK_HIP(hipStreamSynchronize(stream));
    CHECK_APE(ape_model_check(model));
    double msg[S * 25];
    CHECK_HIP(hipMemcpy(msg, msg_dev, sizeof(msg), hipMemcpyDeviceToHost));

    /* ABI 6 / 7: ONE estimator's frame with host buffers -- estimator.py:174-177 (parse_row_to_xx -> add_xx_to_row_hist_and_make_prediction
     * -> msg_from_pred) as one blocking call per frame -- and where the frames' host time went.  The features the rows parse to are
     * written out too (ape_parse_rows): the checker feeds THEM to the oracle's window + model + message. */
    enum { F = 9, RW = 55, DG = 25 + 6 * SMOOTH };
    ape_streams_t* one = NULL;
    CHECK_APE(ape_streams_create(model, 1, T, SMOOTH, &one));
    static float rows[F][RW], feats[F][I], dgram[F][DG];
    lcg_state = 4242u;
    for (int fr = 0; fr < F; ++fr)
        for (int k = 0; k < RW; ++k) rows[fr][k] = lcg_uniform();
    float *rows_dev, *feats_dev;
    CHECK_HIP(hipMalloc((void**)&rows_dev, sizeof(rows)));
    CHECK_HIP(hipMalloc((void**)&feats_dev, sizeof(feats)));
    CHECK_HIP(hipMemcpy(rows_dev, rows, sizeof(rows), hipMemcpyHostToDevice));
    CHECK_APE(ape_parse_rows(APE_PARSE_WATCH_PHONE_POCKET, rows_dev, F, feats_dev, APE_F32, stream));
    CHECK_HIP(hipStreamSynchronize(stream));
    CHECK_HIP(hipMemcpy(feats, feats_dev, sizeof(feats), hipMemcpyDeviceToHost));
    for (int fr = 0; fr < F; ++fr)
        CHECK_APE(ape_streams_frame_host(one, APE_PARSE_WATCH_PHONE_POCKET, rows[fr], APE_FLAG_NORMALIZE_INPUT, dgram[fr], APE_F32, stream));
    ape_frame_stats_t fs;
    float trace[F * 3];
    int32_t n_tr = 0;
    CHECK_APE(ape_streams_frame_stats(one, &fs, trace, F, &n_tr, 0));
    if (fs.frames != F || n_tr != F || fs.recovered != 0) { fprintf(stderr, "frame stats: %llu frames, %d traced, %llu recovered\n",
                                                                      (unsigned long long)fs.frames, (int)n_tr, (unsigned long long)fs.recovered); return 5; }
    for (int fr = 0; fr < F; ++fr)
        if (!(trace[3 * fr] >= 0.0f && trace[3 * fr + 1] >= 0.0f && trace[3 * fr + 2] >= 0.0f)) { fprintf(stderr, "frame trace %d\n", fr); return 5; }
    CHECK_APE(ape_model_check(model));

    FILE* f = fopen(out_path, "wb");
    if (!f) { perror(out_path); return 4; }
    fwrite(y, sizeof(float), (size_t)B * O, f);
    fwrite(est, sizeof(double), (size_t)B * 21, f);
    fwrite(msg, sizeof(double), (size_t)S * 25, f);
    fwrite(feats, sizeof(float), (size_t)F * I, f);
    fwrite(dgram, sizeof(float), (size_t)F * DG, f);
    fclose(f);
    CHECK_APE(ape_streams_destroy(one));
    CHECK_APE(ape_streams_destroy(bank));
    CHECK_APE(ape_model_destroy(model));
    printf("ok: %d windows -> %s (kernel %s)\n", B, out_path, "libape_hip");
    return 0;
}
