// Internal helpers shared by the translation units of libbsq_hip.so (not part of the C ABI).
#pragma once
#include <hip/hip_runtime_api.h>

#include "bsq.h"

namespace bsq_internal {

// Record a thread-local error message (returned by bsq_last_error()) and pass the status through.
bsq_status set_error(bsq_status st, const char *msg);
bsq_status set_hip_error(const char *what, hipError_t e);
// BSQ_NT_STORES=1 selects `global_store ... nt` for the one-hot stream (read once at first use).
bool nontemporal_stores();

}  // namespace bsq_internal
