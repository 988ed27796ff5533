// Internal declarations shared by the translation units of libpgo.so.
// Nothing here is part of the C-ABI (include/pgo.h is).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "pgo.h"

namespace pgo {

// thread-local detail text behind pgo_last_error()
int fail(int status, const std::string& msg);

// ---------------------------------------------------------------- host graph
// Flat-array replacement of ReadG2O's nNodes / nEdgesOdometry / nEdgesClosure /
// nEdgesBogus (reference include/g2o_util.h:174-177).  Edges are kept in
// residual-block order: odometry, closure, bogus (reference main.cpp:95-150).
struct Graph {
  std::vector<int32_t> pose_id;   // Node::index as read from the file
  std::vector<double> pose;       // N x 3
  std::vector<int32_t> ea, eb;    // positions into pose[] (not ids), as the reference indexes nNodes[]
  std::vector<double> meas;       // E x 3
  std::vector<double> info;       // E x 6 (parsed, never used by METHOD 0/1)
  std::vector<uint8_t> kind;      // 0 odometry, 1 closure, 2 bogus
  int32_t n_kind[3] = {0, 0, 0};

  int32_t n_poses() const { return (int32_t)pose_id.size(); }
  int32_t n_edges() const { return (int32_t)ea.size(); }
  // append keeping the odometry|closure|bogus grouping
  void insert_edge(int32_t a, int32_t b, const double m[3], const double inf[6], int kind);
};

// -------------------------------------------------------- shard structure
// Everything the device needs to know about the rows [lo, hi) of J'J owned by
// one rank.  Built once per graph on the host (structure.cpp).
struct ShardStructure {
  int32_t n_poses = 0;       // N (global)
  int32_t world = 1, rank = 0;
  int32_t rows_per_rank = 0; // ceil(N / world); rank r owns [r*rpr, min(N,(r+1)*rpr))
  int32_t lo = 0, hi = 0;    // owned rows
  int32_t n_loc = 0;         // hi - lo

  // local edges = every edge with at least one endpoint in [lo,hi), sorted by
  // (min endpoint, max endpoint, original index)
  int32_t n_edges_local = 0;
  int32_t n_cut = 0;                    // local edges with one endpoint outside [lo,hi)
  std::vector<int32_t> orig_edge;       // local -> caller's edge index
  std::vector<int32_t> ia, ib;          // global pose positions
  std::vector<double> mx, my, mt;       // measurement planes
  std::vector<uint8_t> flags;           // bit0: DCS applies, bit1: cost counted on this rank

  // incidences of the owned rows (row-major):  row i -> [inc_ptr[i-lo], inc_ptr[i-lo+1])
  // each incidence = (local edge, side) ; side 0: row is Edge::a, 1: row is Edge::b
  int64_t n_inc = 0;                    // length of the incidence arrays (with padded tiles: the slots, null incidences included)
  int64_t n_inc_real = 0;               // incidences proper (= 2 x edges between owned rows + 1 x cut edges)
  bool padded = false;                  // pad_tiles_to_slots(): tile t owns the slots [TILE_INC t, TILE_INC (t + 1))
  std::vector<int32_t> inc_ptr;         // n_loc + 1
  std::vector<int32_t> inc_edge;        // (local edge << 1) | side
  std::vector<int32_t> inc_col;         // global pose position of the other endpoint
  std::vector<uint8_t> inc_rowoff;      // row of the incidence - first row of its tile (< TILE_INC)

  // tiles: contiguous row ranges with <= TILE_INC incidences (a row with more
  // incidences forms a tile of its own and is processed in chunks)
  std::vector<int32_t> tile_row;        // n_tiles + 1, LOCAL row index
  int32_t n_tiles() const { return (int32_t)tile_row.size() - 1; }

  // halo of the search direction (world > 1): rows of other ranks that this rank's off-diagonal blocks
  // reference (recv), and rows of this rank that other ranks reference (send); both grouped by peer and
  // sorted by global row, so that peer s's send list to r is exactly r's recv list from s.
  std::vector<int32_t> halo_send_row, halo_recv_row;   // global rows
  std::vector<int64_t> halo_send_off, halo_recv_off;   // world + 1 each, in rows
};

#ifndef PGO_TILE_INC
#define PGO_TILE_INC 256   // (experiment builds may lower it: smaller tiles, more workgroups)
#endif
constexpr int TILE_INC = PGO_TILE_INC;

inline int32_t rows_per_rank(int32_t n_poses, int world, int row_align) {
  const int64_t rpr = ((int64_t)n_poses + world - 1) / world;
  return (int32_t)(((rpr + row_align - 1) / row_align) * row_align);
}

int build_shard_structure(int32_t n_poses, int32_t n_edges, const int32_t* ia, const int32_t* ib,
                          const double* meas, const uint8_t* kind, int method, int world, int rank, int row_align,
                          ShardStructure* out, const std::vector<int32_t>* tile_breaks = nullptr);

// processing order of the row tiles for K3 (structure.cpp): order[k] = tile that takes the k-th turn
void compute_tile_order(const ShardStructure& S, std::vector<int32_t>* order);
bool pad_tiles_to_slots(ShardStructure* S);

// locality ordering (structure.cpp): perm[i] = new position of pose i
int compute_pose_order(int32_t n_poses, int32_t n_edges, const int32_t* ia, const int32_t* ib, int32_t segment,
                       std::vector<int32_t>* perm);
constexpr int ORDER_SEGMENT = 64;
// the same through a small process-wide cache keyed by a hash of the edge list (repeated handles on one graph)
int cached_pose_order(int32_t n_poses, int32_t n_edges, const int32_t* ia, const int32_t* ib, int32_t segment,
                      std::vector<int32_t>* perm);

// poses per block of the block-Jacobi preconditioner for an option value (0 = auto)
inline int resolve_block_poses(int opt_value, int32_t n_poses) {
  int b = opt_value;
  if (b <= 0) b = (n_poses <= 8192) ? 32 : 4;  // measured: 1M poses 32.0 (B=1) / 37.7 (B=4) / 34.0 (B=8) GN it/s
  return b > 32 ? 32 : b;
}

// segment length of the chain (block-tridiagonal) preconditioner for an option value: 0 = off, -1 = auto when the block
// size is also left to auto:
//   * more than 50000 poses: 64.  Measured GN it/s, chain-64 vs dense blocks: 1M poses 47.6 vs 36.4, 100k 328 vs 303,
//     10k 401 vs 594 (a wavefront covers 256 poses: too few waves on mid-size graphs);
//   * up to 8192 poses (launch-bound, exact mode): 256 if the graph is chain-like -- at most 5 % as many SHORT-RANGE
//     non-consecutive edges (2 <= |a-b| < 32) as poses -- else off (dense 32-pose blocks, which capture such edges; the
//     tridiagonal chain does not).  Measured, 50 LM iterations, chain-256 vs B=32: INTEL 0.40 vs 0.48 s, MIT METHOD 1
//     0.21 vs 1.42 s, CSAIL 0.44 vs 0.46, FR079 0.41 vs 0.45; FRH 2.07 vs 0.46 and M3500 2.25 vs 1.20 (short-range rich).
inline int resolve_chain_len(int opt_value, int opt_block_poses, int32_t n_poses, int32_t n_edges = 0,
                             const int32_t* ia = nullptr, const int32_t* ib = nullptr) {
  if (opt_value >= 0) return opt_value;
  if (opt_block_poses > 0) return 0;
  if (n_poses > 50000) return 64;
  if (n_poses <= 8192 && n_poses >= 512 && ia && ib) {
    int64_t short_range = 0;
    for (int32_t e = 0; e < n_edges; ++e) {
      const int32_t d = ia[e] > ib[e] ? ia[e] - ib[e] : ib[e] - ia[e];
      short_range += (d >= 2 && d < 32);
    }
    if (20 * short_range <= (int64_t)n_poses) return 256;
  }
  return 0;
}

}  // namespace pgo

struct pgo_graph {
  pgo::Graph g;
};
