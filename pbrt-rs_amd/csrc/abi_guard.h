// abi_guard.h — nothing is thrown across the C ABI ("int status returns, no exceptions or panics across the
// boundary"; the reference's own error channel is panic!, which a C caller cannot catch either). Every status-returning entry
// point is a function-try-block ending in PB_ABI_CATCH: a host allocation that fails (std::vector / std::string growing while a
// tree is built or a scene re-laid out) becomes PBRT_HIP_ERR_OOM, anything else PBRT_HIP_ERR_INVALID. Locks and device buffers
// are RAII objects: unwinding releases them.
#pragma once
#include <new>

#include "../../include/pbrt_hip.h"

#define PB_ABI_CATCH                                  \
    catch (const std::bad_alloc&) {                   \
        return PBRT_HIP_ERR_OOM;                      \
    }                                                 \
    catch (...) {                                     \
        return PBRT_HIP_ERR_INVALID;                  \
    }
