// Communicators for the pose-id-range sharded solve: one process per GPU.
//
//   rccl : RCCL over xGMI (production).  Three collectives only, all on the
//          solver's stream: all-reduce(sum|max) of a few doubles (CG dot products,
//          LM scalars) and an in-place all-gather of the search direction / pose
//          update (3N doubles).
//   shm  : host-staged POSIX shared memory.  TEST backend: lets several ranks share
//          one GPU (RCCL refuses duplicate devices) so that the sharded numerics
//          can be checked on a 1-GPU box.  Rank-ordered sums => deterministic.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "comm.h"
#include "pgo_internal.h"

long long knob(const char* name);   // test hooks (solver_abi.hip)

namespace pgo {

// ------------------------------------------------------------------ RCCL
struct RcclComm final : pgo_comm {
  ncclComm_t comm = nullptr;
  ~RcclComm() override {
    if (comm) ncclCommDestroy(comm);
  }
  bool capturable() const override { return true; }
  int allreduce(double* dev, int n, bool is_max, hipStream_t s) override {
    ncclResult_t r = ncclAllReduce(dev, dev, (size_t)n, ncclDouble, is_max ? ncclMax : ncclSum, comm, s);
    if (r != ncclSuccess) return fail(PGO_ERR_COMM, std::string("ncclAllReduce: ") + ncclGetErrorString(r));
    return PGO_OK;
  }
  int allgather_inplace(double* base, int64_t count_per_rank, hipStream_t s) override {
    ncclResult_t r = ncclAllGather(base + (int64_t)rank * count_per_rank, base, (size_t)count_per_rank, ncclDouble, comm, s);
    if (r != ncclSuccess) return fail(PGO_ERR_COMM, std::string("ncclAllGather: ") + ncclGetErrorString(r));
    return PGO_OK;
  }
  int exchange(const double* sendbuf, const int64_t* send_off, double* recvbuf, const int64_t* recv_off,
               hipStream_t s) override {
    ncclResult_t r = ncclGroupStart();
    for (int peer = 0; peer < world && r == ncclSuccess; ++peer) {
      if (peer == rank) continue;
      const int64_t ns = send_off[peer + 1] - send_off[peer], nr = recv_off[peer + 1] - recv_off[peer];
      if (ns > 0) r = ncclSend(sendbuf + send_off[peer], (size_t)ns, ncclDouble, peer, comm, s);
      if (nr > 0 && r == ncclSuccess) r = ncclRecv(recvbuf + recv_off[peer], (size_t)nr, ncclDouble, peer, comm, s);
    }
    ncclResult_t r2 = ncclGroupEnd();
    if (r != ncclSuccess || r2 != ncclSuccess)
      return fail(PGO_ERR_COMM, std::string("ncclSend/ncclRecv group: ") + ncclGetErrorString(r != ncclSuccess ? r : r2));
    return PGO_OK;
  }
};

// ------------------------------------------------------------------- shm
struct ShmHeader {
  std::atomic<int> arrived;
  std::atomic<int> generation;
  std::atomic<int> attached;
  int world;
  int64_t slot_bytes;
};

struct ShmComm final : pgo_comm {
  std::string name;
  ShmHeader* hdr = nullptr;
  char* data = nullptr;
  size_t map_bytes = 0;
  int64_t slot_bytes = 0;
  std::vector<double> tmp;

  ~ShmComm() override {
    if (hdr) munmap((void*)hdr, map_bytes);
    if (rank == 0) shm_unlink(name.c_str());
  }
  int barrier() {
    const long long kt = knob("shm_timeout_s");       // (test hook: campaigns over many small cases shorten the wait for a dead peer)
    const long long timeout_s = kt > 0 ? kt : 120;
    const int gen = hdr->generation.load(std::memory_order_acquire);
    if (hdr->arrived.fetch_add(1, std::memory_order_acq_rel) == world - 1) {
      hdr->arrived.store(0, std::memory_order_relaxed);
      hdr->generation.fetch_add(1, std::memory_order_acq_rel);
      return PGO_OK;
    }
    auto t0 = std::chrono::steady_clock::now();
    int spins = 0;
    while (hdr->generation.load(std::memory_order_acquire) == gen) {
      if (++spins > 256) {
        std::this_thread::sleep_for(std::chrono::microseconds(20));
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(timeout_s))
          return fail(PGO_ERR_COMM, "shm barrier timeout (a peer died?)");
      }
    }
    return PGO_OK;
  }
  char* slot(int r) { return data + (int64_t)r * slot_bytes; }

  int allreduce(double* dev, int n, bool is_max, hipStream_t s) override {
    if ((int64_t)n * 8 > slot_bytes) return fail(PGO_ERR_COMM, "shm allreduce larger than a slot");
    if (hipMemcpyAsync(slot(rank), dev, (size_t)n * 8, hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess)
      return fail(PGO_ERR_HIP, "shm allreduce D2H");
    int st = barrier();
    if (st) return st;
    tmp.assign((size_t)n, 0.0);
    for (int r = 0; r < world; ++r) {  // rank order: same sum on every rank
      const double* src = (const double*)slot(r);
      for (int i = 0; i < n; ++i) tmp[i] = is_max ? (r == 0 ? src[i] : (src[i] > tmp[i] ? src[i] : tmp[i])) : tmp[i] + src[i];
    }
    st = barrier();  // everybody has read the slots
    if (st) return st;
    if (hipMemcpyAsync(dev, tmp.data(), (size_t)n * 8, hipMemcpyHostToDevice, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess)
      return fail(PGO_ERR_HIP, "shm allreduce H2D");
    return PGO_OK;
  }
  int allgather_inplace(double* base, int64_t cpr, hipStream_t s) override {
    const int64_t chunk = slot_bytes / 8;
    for (int64_t off = 0; off < cpr; off += chunk) {
      const int64_t m = (cpr - off < chunk) ? cpr - off : chunk;
      if (hipMemcpyAsync(slot(rank), base + (int64_t)rank * cpr + off, (size_t)m * 8, hipMemcpyDeviceToHost, s) != hipSuccess ||
          hipStreamSynchronize(s) != hipSuccess)
        return fail(PGO_ERR_HIP, "shm allgather D2H");
      int st = barrier();
      if (st) return st;
      for (int r = 0; r < world; ++r) {
        if (r == rank) continue;
        if (hipMemcpyAsync(base + (int64_t)r * cpr + off, slot(r), (size_t)m * 8, hipMemcpyHostToDevice, s) != hipSuccess)
          return fail(PGO_ERR_HIP, "shm allgather H2D");
      }
      if (hipStreamSynchronize(s) != hipSuccess) return fail(PGO_ERR_HIP, "shm allgather sync");
      st = barrier();
      if (st) return st;
    }
    return PGO_OK;
  }
  int exchange(const double* sendbuf, const int64_t* send_off, double* recvbuf, const int64_t* recv_off,
               hipStream_t s) override {
    // slot layout: [world + 1 offsets (int64)] [all outgoing segments]
    const int64_t hdr_bytes = (int64_t)(world + 1) * 8, total = send_off[world];
    if (hdr_bytes + total * 8 > slot_bytes) return fail(PGO_ERR_COMM, "shm exchange larger than a slot");
    memcpy(slot(rank), send_off, (size_t)hdr_bytes);
    if (total > 0 && (hipMemcpyAsync(slot(rank) + hdr_bytes, sendbuf, (size_t)total * 8, hipMemcpyDeviceToHost, s) != hipSuccess ||
                      hipStreamSynchronize(s) != hipSuccess))
      return fail(PGO_ERR_HIP, "shm exchange D2H");
    int st = barrier();
    if (st) return st;
    for (int peer = 0; peer < world; ++peer) {
      if (peer == rank) continue;
      const int64_t* poff = (const int64_t*)slot(peer);
      const int64_t n = poff[rank + 1] - poff[rank];
      if (n != recv_off[peer + 1] - recv_off[peer]) return fail(PGO_ERR_COMM, "shm exchange: send/recv counts disagree");
      if (n > 0 && hipMemcpyAsync(recvbuf + recv_off[peer], slot(peer) + hdr_bytes + poff[rank] * 8, (size_t)n * 8,
                                  hipMemcpyHostToDevice, s) != hipSuccess)
        return fail(PGO_ERR_HIP, "shm exchange H2D");
    }
    if (hipStreamSynchronize(s) != hipSuccess) return fail(PGO_ERR_HIP, "shm exchange sync");
    return barrier();
  }
};

}  // namespace pgo

using pgo::fail;

extern "C" {

int pgo_comm_unique_id(uint8_t id[PGO_COMM_ID_BYTES]) {
  static_assert(PGO_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
  if (!id) return fail(PGO_ERR_INVALID_ARG, "pgo_comm_unique_id: null");
  ncclUniqueId u;
  ncclResult_t r = ncclGetUniqueId(&u);
  if (r != ncclSuccess) return fail(PGO_ERR_COMM, std::string("ncclGetUniqueId: ") + ncclGetErrorString(r));
  memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
  return PGO_OK;
}

int pgo_comm_create_rccl(const uint8_t id[PGO_COMM_ID_BYTES], int rank, int world, int device, pgo_comm** out) {
  if (!id || !out || world < 1 || rank < 0 || rank >= world) return fail(PGO_ERR_INVALID_ARG, "pgo_comm_create_rccl: bad argument");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(PGO_ERR_NO_DEVICE, "no HIP device");
  if (device < 0 || device >= ndev) return fail(PGO_ERR_INVALID_ARG, "device index out of range");
  if (hipSetDevice(device) != hipSuccess) return fail(PGO_ERR_HIP, "hipSetDevice");
  auto* c = new pgo::RcclComm;
  c->rank = rank;
  c->world = world;
  c->device = device;
  ncclUniqueId u;
  memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
  ncclResult_t r = ncclCommInitRank(&c->comm, world, u, rank);
  if (r != ncclSuccess) {
    c->comm = nullptr;
    delete c;
    return fail(PGO_ERR_COMM, std::string("ncclCommInitRank: ") + ncclGetErrorString(r));
  }
  *out = c;
  return PGO_OK;
}

int pgo_comm_create_shm(const char* name, int rank, int world, int device, pgo_comm** out) {
  if (!name || !out || world < 1 || rank < 0 || rank >= world) return fail(PGO_ERR_INVALID_ARG, "pgo_comm_create_shm: bad argument");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(PGO_ERR_NO_DEVICE, "no HIP device");
  if (device < 0 || device >= ndev) return fail(PGO_ERR_INVALID_ARG, "device index out of range");
  if (hipSetDevice(device) != hipSuccess) return fail(PGO_ERR_HIP, "hipSetDevice");
  auto* c = new pgo::ShmComm;
  c->rank = rank;
  c->world = world;
  c->device = device;
  c->name = std::string("/") + name;
  c->slot_bytes = 8 << 20;
  const size_t hdr_bytes = 4096;
  c->map_bytes = hdr_bytes + (size_t)world * (size_t)c->slot_bytes;
  int fd = -1;
  if (rank == 0) {
    shm_unlink(c->name.c_str());
    fd = shm_open(c->name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)c->map_bytes) != 0) {
      if (fd >= 0) close(fd);
      delete c;
      return fail(PGO_ERR_COMM, "shm_open/ftruncate failed");
    }
  } else {
    for (int tries = 0; tries < 6000 && fd < 0; ++tries) {  // wait for rank 0 (<= 60 s)
      fd = shm_open(c->name.c_str(), O_RDWR, 0600);
      if (fd >= 0) {
        struct stat sb;
        if (fstat(fd, &sb) != 0 || (size_t)sb.st_size < c->map_bytes) {
          close(fd);
          fd = -1;
        }
      }
      if (fd < 0) std::this_thread::sleep_for(std::chrono::milliseconds(10));
    }
    if (fd < 0) {
      delete c;
      return fail(PGO_ERR_COMM, "shm segment never appeared");
    }
  }
  void* m = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (m == MAP_FAILED) {
    delete c;
    return fail(PGO_ERR_COMM, "mmap failed");
  }
  c->hdr = (pgo::ShmHeader*)m;
  c->data = (char*)m + hdr_bytes;
  if (rank == 0) {
    c->hdr->world = world;
    c->hdr->slot_bytes = c->slot_bytes;
    c->hdr->arrived.store(0);
    c->hdr->generation.store(0);
    c->hdr->attached.store(1, std::memory_order_release);
  } else {
    for (int tries = 0; tries < 6000 && c->hdr->attached.load(std::memory_order_acquire) == 0; ++tries)
      std::this_thread::sleep_for(std::chrono::milliseconds(10));
    if (c->hdr->attached.load() == 0 || c->hdr->world != world) {
      delete c;
      return fail(PGO_ERR_COMM, "shm header not initialised or world mismatch");
    }
    c->hdr->attached.fetch_add(1);
  }
  int st = c->barrier();
  if (st) {
    delete c;
    return st;
  }
  *out = c;
  return PGO_OK;
}

void pgo_comm_destroy(pgo_comm* c) { delete c; }

}  // extern "C"
