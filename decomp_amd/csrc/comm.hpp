// The handle's RCCL communicator (the ONE exchange of the data-parallel path, the
// all-reduce of the [K, F+K] statistics, runs inside the library on the handle's own stream).
// librccl is bound at run time (dlopen: the copy the process already holds -- torch ships one -- or the
// system one), so the library itself has no link-time dependency on it and loads on hosts without RCCL.
#pragma once
#include <stddef.h>

#include "handle.hpp"

namespace dcp {

enum { COMM_F32 = 0, COMM_F64 = 1 };

inline bool comm_active(const dcp_handle* h) { return h->comm != nullptr || h->comm_ext != nullptr; }

// In-place sum over the ranks of the handle's communicator, enqueued on h->stream (no host wait).
int comm_allreduce_sum(dcp_handle* h, void* buf, size_t count, int dtype);

// dcp_destroy: tear the communicator down (no-op without one).
void comm_release(dcp_handle* h);

}  // namespace dcp
