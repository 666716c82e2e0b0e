// Internal helpers shared by the translation units of libbsq_hip.so (not part of the C ABI).
#pragma once
#include <hip/hip_runtime_api.h>

#include "bsq.h"

namespace bsq_internal {

// Record a thread-local error message (returned by bsq_last_error()) and pass the status through.
bsq_status set_error(bsq_status st, const char *msg);
bsq_status set_hip_error(const char *what, hipError_t e);
// Tuning / diagnostic knobs (bsq_tuning_set, or environment BSQ_<NAME> read at first use):
//   nt_stores   1: `global_store ... nt` for the one-hot stream          (default 1)
//   onehot_path 0: automatic, 1: tiled kernel, 2: two-pass (tokens + expansion), 3: chunk-owner kernel
//   expand_cpw  chunks per wave of the chunk kernels (default 1)
//   onehot_tb   0: automatic, else force 64 / 128 / 256 sequences per tile
//   tile_order  0: sequence-tile index fastest, 1: position-tile index fastest
//   fill_mode   access pattern of bsq_fill_device (write-bandwidth experiments)
int tuning(const char *name);
bool set_tuning(const char *name, int value);
inline bool nontemporal_stores() { return tuning("nt_stores") != 0; }

// Stream-ordered scratch (hipMallocAsync from a pool that keeps its memory between calls).
bsq_status workspace_acquire(size_t nbytes, hipStream_t stream, void **ptr);
void workspace_release(void *ptr, hipStream_t stream);

}  // namespace bsq_internal
