// Test hooks: linked into lib/diag/libape_hip_testhooks.so only (csrc/Makefile, target `hooks`), never into the product library.
#include "ape_model.h"

#define APE_DBG_TRY(expr) do { if ((expr) != hipSuccess) return APE_ERR_HIP; } while (0)

extern "C" {

// internal (not in the public header): overwrite one control word of the cluster kernels -- 0 status, 1 ticket,
// 2 departure counter, 3 launch number of the latency kernel, 4 / 5 status word / a class ticket of the MLP pipeline, 6 the class-0 ticket of
// the cluster kernels that form their clusters within block-index classes, 7 the launch number of the Monte-Carlo latency kernel, 8 the launch
// number of the level-synchronous kernel (lstm_level16.hip) -- so that tests
// can stage the state an aborted launch leaves behind
int ape_debug_poke(ape_model_t* m, int which, unsigned value) {
    if (m && m->ffp_ok && (which == 4 || which == 5)) {          // the MLP pipeline's status word / first class ticket
        APE_DBG_TRY(hipSetDevice(m->dims.device));
        APE_DBG_TRY(hipDeviceSynchronize());
        APE_DBG_TRY(hipMemcpy(m->ffp_ctl + (which == 4 ? 8 * 16 : 0), &value, sizeof(value), hipMemcpyHostToDevice));
        return APE_OK;
    }
    if (m && m->cluster_ok && which == 6) {
        APE_DBG_TRY(hipSetDevice(m->dims.device));
        APE_DBG_TRY(hipDeviceSynchronize());
        APE_DBG_TRY(hipMemcpy(m->xcc_slots + 64, &value, sizeof(value), hipMemcpyHostToDevice));
        return APE_OK;
    }
    if (m && m->mcs_ok && which == 7) {          // the Monte-Carlo latency kernel's launch number (upper bits of its granule tags)
        APE_DBG_TRY(hipSetDevice(m->dims.device));
        APE_DBG_TRY(hipDeviceSynchronize());
        APE_DBG_TRY(hipMemcpy(m->gxm, &value, sizeof(value), hipMemcpyHostToDevice));
        return APE_OK;
    }
    if (m && m->lv16_ok && which == 8) {         // the level-synchronous kernel's launch number (upper bits of its granule tags)
        APE_DBG_TRY(hipSetDevice(m->dims.device));
        APE_DBG_TRY(hipDeviceSynchronize());
        APE_DBG_TRY(hipMemcpy(m->gx16, &value, sizeof(value), hipMemcpyHostToDevice));
        return APE_OK;
    }
    if (!m || !m->cluster_ok || which < 0 || which > 3) return APE_ERR_INVALID_ARG;
    APE_DBG_TRY(hipSetDevice(m->dims.device));
    APE_DBG_TRY(hipDeviceSynchronize());
    unsigned* status = m->xflags + m->xflag_bytes / sizeof(unsigned);
    // (3: the latency kernel's launch number, the upper bits of its granule tags -- to stage the 20-bit wrap)
    unsigned* word = which == 0 ? status : (which == 1 ? status - 4 : which == 2 ? status - 3 : reinterpret_cast<unsigned*>(m->hxs));
    APE_DBG_TRY(hipMemcpy(word, &value, sizeof(value), hipMemcpyHostToDevice));
    return APE_OK;
}


// shrink the chunk of sample rows a Monte-Carlo bank's weight-stationary route handles per launch (a multiple of 32, never
// above what the bank's workspaces were sized for): lets a test run the multi-chunk path at a size that otherwise fits one
int ape_debug_set_chunk_rows(ape_streams_t* b, int rows) {
    if (!b || !(b->up32 || b->up128) || rows < 32 || rows % 32 != 0 || rows > b->chunk_rows) return APE_ERR_INVALID_ARG;
    b->chunk_rows = rows;
    return APE_OK;
}

// injected dropout multipliers for a Monte-Carlo bank, [L-1, S * n_mc, T, H] float32 on the device (NULL: back to the Philox draws): the
// bank's routes against the oracle's masked loop under the SAME masks (the product library draws its masks itself)
int ape_debug_set_bank_masks(ape_streams_t* b, const float* masks_dev) {
    if (!b) return APE_ERR_INVALID_ARG;
    b->inj_masks = masks_dev;
    return APE_OK;
}

// the normalised NN targets of the bank's newest step, [S * n_mc, O] float32 -> host
int ape_debug_bank_targets(ape_streams_t* b, float* out_host) {
    if (!b || !out_host || !b->y_new) return APE_ERR_INVALID_ARG;
    APE_DBG_TRY(hipSetDevice(b->model->dims.device));
    APE_DBG_TRY(hipDeviceSynchronize());
    APE_DBG_TRY(hipMemcpy(out_host, b->y_new, (size_t)b->S * b->n_mc * b->model->dims.output_size * sizeof(float), hipMemcpyDeviceToHost));
    return APE_OK;
}

// one of the bank's device buffers -> host, for localising a wrong frame: 0 the feature ring [S, n_mc, T, I], 1 layer 0's input tiles
// (fragment order, ape_lower32_xfrag_bytes), 2 layer 0's output sequence (fragment order), 3 the masked input of the last chunk of launch B.
// `bytes` = what the caller's buffer holds; the copy is the smaller of that and the buffer (returned in *copied)
int ape_debug_bank_buffer(ape_streams_t* b, int which, void* out_host, size_t bytes, size_t* copied) {
    if (!b || !out_host || which < 0 || which > 3) return APE_ERR_INVALID_ARG;
    const int I = b->model->dims.input_size;
    const size_t tiles = (size_t)(b->S + 31) / 32, ctiles = (size_t)(b->chunk_rows + 31) / 32;
    const void* src = which == 0 ? (const void*)b->xring : which == 1 ? (const void*)b->xfrag0 : which == 2 ? (const void*)b->hfrag : (const void*)b->xfrag;
    size_t have = which == 0 ? (size_t)b->S * b->n_mc * b->T * I * sizeof(float) : which == 1 ? tiles * b->T * 4096 : which == 2 ? tiles * b->T * 32768 : ctiles * b->T * 32768;
    if (!src) return APE_ERR_NOT_READY;
    if (have > bytes) have = bytes;
    APE_DBG_TRY(hipSetDevice(b->model->dims.device));
    APE_DBG_TRY(hipDeviceSynchronize());
    APE_DBG_TRY(hipMemcpy(out_host, src, have, hipMemcpyDeviceToHost));
    if (copied) *copied = have;
    return APE_OK;
}

}  // extern "C"
