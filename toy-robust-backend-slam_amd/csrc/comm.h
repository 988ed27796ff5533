// Communicator interface used by the solver (internal; the C-ABI sees an opaque pgo_comm*).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct pgo_comm {
  int rank = 0, world = 1, device = 0;
  virtual ~pgo_comm() {}
  // in-place all-reduce of n doubles in device memory (sum or max), ordered on `s`
  virtual int allreduce(double* dev, int n, bool is_max, hipStream_t s) = 0;
  // in-place all-gather: rank r's `count_per_rank` doubles live at base + r*count_per_rank
  virtual int allgather_inplace(double* base, int64_t count_per_rank, hipStream_t s) = 0;
};
