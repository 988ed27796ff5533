// Communicator interface used by the solver (internal; the C-ABI sees an opaque pgo_comm*).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct pgo_comm {
  int rank = 0, world = 1, device = 0;
  virtual ~pgo_comm() {}
  // true when the collectives are pure stream work and may be recorded into a hipGraph (RCCL); the host-staged test
  // back-end blocks on the host inside every call and may not
  virtual bool capturable() const { return false; }
  // in-place all-reduce of n doubles in device memory (sum or max), ordered on `s`
  virtual int allreduce(double* dev, int n, bool is_max, hipStream_t s) = 0;
  // in-place all-gather: rank r's `count_per_rank` doubles live at base + r*count_per_rank
  virtual int allgather_inplace(double* base, int64_t count_per_rank, hipStream_t s) = 0;
  // all-to-all-v of doubles in device memory: segment [send_off[r], send_off[r+1]) of sendbuf goes to rank r,
  // segment [recv_off[r], recv_off[r+1]) of recvbuf comes from rank r (offsets in doubles, world + 1 entries,
  // host memory; the own-rank segments are empty)
  virtual int exchange(const double* sendbuf, const int64_t* send_off, double* recvbuf, const int64_t* recv_off,
                       hipStream_t s) = 0;
};
