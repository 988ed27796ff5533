// Symbolic phase, host side, once per graph: which rows of J'J a rank owns, which
// edges it must evaluate, the row -> incidence lists that drive the assembly and
// SpMV kernels, and the tiling of rows into workgroup-sized pieces.
//
// Sharding rule (north_star: "shards by pose-id range"):
//   rank r owns rows [r*rpr, min(N,(r+1)*rpr)), rpr = ceil(N/world) rounded up to a multiple of
//   `row_align` (the preconditioner's pose-block size: shard boundaries then coincide with block
//   boundaries, so the preconditioner -- and with it the PCG iterates -- do not depend on the world size).
//   A rank evaluates EVERY edge touching one of its rows (cut edges are evaluated
//   on both owners - 81 bytes of input per edge - instead of exchanging 112-byte
//   Jacobian records), so assembly needs no communication at all.  An edge's cost
//   is counted on the rank that owns its first endpoint Edge::a.
#include <algorithm>
#include <mutex>
#include <numeric>
#include <thread>
#include <queue>
#include <unordered_set>

#include "pgo_internal.h"

namespace pgo {

// [0, n) in contiguous ranges over up to 8 host threads (large inputs only); f(begin, end)
template <class F>
static void parallel_ranges(int64_t n, F f) {
  const unsigned hw = std::thread::hardware_concurrency();
  const int nt = (n < (1 << 16)) ? 1 : (int)std::min<unsigned>(8, hw ? hw : 1);
  if (nt <= 1) {
    f(0, n);
    return;
  }
  std::vector<std::thread> pool;
  for (int t = 0; t < nt; ++t) pool.emplace_back([=] { f(n * t / nt, n * (t + 1) / nt); });
  for (auto& th : pool) th.join();
}

// Locality ordering of the poses (see pgo_pose_order in pgo.h).
int compute_pose_order(int32_t N, int32_t E, const int32_t* ia, const int32_t* ib, int32_t L, std::vector<int32_t>* perm) {
  if (N <= 0 || E < 0 || L < 1) return fail(PGO_ERR_INVALID_ARG, "compute_pose_order: bad sizes");
  perm->resize(N);
  const int32_t n_full = N / L;  // full segments; a short tail segment keeps its place at the end
  if (n_full < 3) {
    std::iota(perm->begin(), perm->end(), 0);
    return PGO_OK;
  }
  // supported loop edges -> edges of the segment graph.  "Is there an edge {x, y}?" is answered from a symmetric
  // adjacency in CSR form with sorted rows (two counting passes + a per-row sort; a hash set of the 4M pairs of the
  // 1M-pose graph took 2.1 s here, on every rank)
  std::vector<int64_t> aptr((size_t)N + 1, 0);
  for (int32_t e = 0; e < E; ++e) {
    aptr[(size_t)ia[e] + 1]++;
    aptr[(size_t)ib[e] + 1]++;
  }
  for (int32_t i = 0; i < N; ++i) aptr[(size_t)i + 1] += aptr[i];
  std::vector<int32_t> adj_col((size_t)aptr[N]);
  {
    std::vector<int64_t> fill(aptr.begin(), aptr.end() - 1);
    for (int32_t e = 0; e < E; ++e) {
      adj_col[(size_t)fill[ia[e]]++] = ib[e];
      adj_col[(size_t)fill[ib[e]]++] = ia[e];
    }
    parallel_ranges(N, [&](int64_t b, int64_t e) {
      for (int64_t i = b; i < e; ++i) std::sort(adj_col.begin() + aptr[i], adj_col.begin() + aptr[(size_t)i + 1]);
    });
  }
  auto has_edge = [&](int64_t x, int64_t y) {
    return std::binary_search(adj_col.begin() + aptr[x], adj_col.begin() + aptr[x + 1], (int32_t)y);
  };
  std::vector<std::pair<int32_t, int32_t>> seg_edges;
  {
    std::mutex mu;
    parallel_ranges(E, [&](int64_t e0, int64_t e1) {
      std::vector<std::pair<int32_t, int32_t>> mine;
      for (int64_t e = e0; e < e1; ++e) {
        const int64_t a = ia[e], b = ib[e];
        const int32_t sa = (int32_t)(a / L), sb = (int32_t)(b / L);
        if (sa == sb || sa >= n_full || sb >= n_full) continue;
        bool ok = std::llabs(a - b) <= 1;
        for (int da = -1; da <= 1 && !ok; ++da)
          for (int db = -1; db <= 1 && !ok; ++db) {
            if (da == 0 && db == 0) continue;
            const int64_t x = a + da, y = b + db;
            if (x < 0 || y < 0 || x >= N || y >= N || x == y) continue;
            ok = has_edge(x, y);
          }
        if (ok) mine.emplace_back(std::min(sa, sb), std::max(sa, sb));
      }
      std::lock_guard<std::mutex> lk(mu);
      seg_edges.insert(seg_edges.end(), mine.begin(), mine.end());   // (sorted and made unique below: the order does not matter)
    });
  }
  std::sort(seg_edges.begin(), seg_edges.end());
  seg_edges.erase(std::unique(seg_edges.begin(), seg_edges.end()), seg_edges.end());
  std::vector<int32_t> ptr((size_t)n_full + 1, 0), adj(seg_edges.size() * 2);
  for (auto& pr : seg_edges) {
    ptr[pr.first + 1]++;
    ptr[pr.second + 1]++;
  }
  for (int32_t i = 0; i < n_full; ++i) ptr[i + 1] += ptr[i];
  {
    std::vector<int32_t> fill(ptr.begin(), ptr.end() - 1);
    for (auto& pr : seg_edges) {
      adj[fill[pr.first]++] = pr.second;
      adj[fill[pr.second]++] = pr.first;
    }
  }
  auto deg = [&](int32_t v) { return ptr[v + 1] - ptr[v]; };
  // Cuthill-McKee per connected component, components entered at their lowest-degree segment in index order
  std::vector<int32_t> order;
  order.reserve(n_full);
  std::vector<char> seen(n_full, 0);
  std::vector<int32_t> by_deg(n_full);
  std::iota(by_deg.begin(), by_deg.end(), 0);
  std::stable_sort(by_deg.begin(), by_deg.end(), [&](int32_t x, int32_t y) { return deg(x) < deg(y); });
  std::vector<int32_t> nb;
  for (int32_t start : by_deg) {
    if (seen[start]) continue;
    size_t head = order.size();
    order.push_back(start);
    seen[start] = 1;
    while (head < order.size()) {
      const int32_t v = order[head++];
      nb.clear();
      for (int32_t q = ptr[v]; q < ptr[v + 1]; ++q)
        if (!seen[adj[q]]) {
          seen[adj[q]] = 1;
          nb.push_back(adj[q]);
        }
      std::sort(nb.begin(), nb.end(), [&](int32_t x, int32_t y) { return deg(x) != deg(y) ? deg(x) < deg(y) : x < y; });
      order.insert(order.end(), nb.begin(), nb.end());
    }
  }
  std::reverse(order.begin(), order.end());
  std::vector<int32_t> start_of(n_full);
  for (int32_t k = 0; k < n_full; ++k) start_of[order[k]] = k * L;
  for (int32_t i = 0; i < N; ++i) {
    const int32_t sgm = i / L;
    (*perm)[i] = sgm < n_full ? start_of[sgm] + (i - sgm * L) : i;
  }
  return PGO_OK;
}

int cached_pose_order(int32_t N, int32_t E, const int32_t* ia, const int32_t* ib, int32_t L, std::vector<int32_t>* perm) {
  struct Entry {
    int32_t n = 0, e = 0, l = 0;
    uint64_t h = 0;
    std::vector<int32_t> perm;
  };
  static std::mutex mu;
  static Entry cache[2];
  static int next = 0;
  uint64_t h = 1469598103934665603ull;   // FNV-1a over the endpoint words
  for (int32_t e = 0; e < E; ++e) {
    h = (h ^ (uint32_t)ia[e]) * 1099511628211ull;
    h = (h ^ (uint32_t)ib[e]) * 1099511628211ull;
  }
  {
    std::lock_guard<std::mutex> lk(mu);
    for (const Entry& c : cache)
      if (c.n == N && c.e == E && c.l == L && c.h == h && (int32_t)c.perm.size() == N) {
        *perm = c.perm;
        return PGO_OK;
      }
  }
  int st = compute_pose_order(N, E, ia, ib, L, perm);
  if (st != PGO_OK) return st;
  std::lock_guard<std::mutex> lk(mu);
  Entry& c = cache[next];
  next ^= 1;
  c.n = N; c.e = E; c.l = L; c.h = h;
  c.perm = *perm;
  return PGO_OK;
}

// Halo of the gather vector for world > 1 (needs the WHOLE edge list: what the peers reference of this
// rank's rows).  Both lists are grouped by peer and sorted by global row, so that rank s's send list to r is
// exactly r's receive list from s.
static void build_halo_lists(int32_t E, const int32_t* ia, const int32_t* ib, ShardStructure* S) {
  const int world = S->world, rank = S->rank;
  S->halo_send_off.assign((size_t)world + 1, 0);
  S->halo_recv_off.assign((size_t)world + 1, 0);
  S->halo_send_row.clear();
  S->halo_recv_row.clear();
  if (world <= 1) return;
  const int32_t rpr = S->rows_per_rank;
  std::vector<std::vector<int32_t>> snd(world), rcv(world);
  for (int32_t e = 0; e < E; ++e) {
    const int oa = ia[e] / rpr, ob = ib[e] / rpr;
    if (oa == ob) continue;
    if (oa == rank) {
      snd[ob].push_back(ia[e]);  // peer ob's row ib[e] has a block in column ia[e]
      rcv[ob].push_back(ib[e]);
    }
    if (ob == rank) {
      snd[oa].push_back(ib[e]);
      rcv[oa].push_back(ia[e]);
    }
  }
  for (int s = 0; s < world; ++s) {
    for (auto* v : {&snd[s], &rcv[s]}) {
      std::sort(v->begin(), v->end());
      v->erase(std::unique(v->begin(), v->end()), v->end());
    }
    S->halo_send_row.insert(S->halo_send_row.end(), snd[s].begin(), snd[s].end());
    S->halo_recv_row.insert(S->halo_recv_row.end(), rcv[s].begin(), rcv[s].end());
    S->halo_send_off[s + 1] = (int64_t)S->halo_send_row.size();
    S->halo_recv_off[s + 1] = (int64_t)S->halo_recv_row.size();
  }
}

int build_shard_structure(int32_t N, int32_t E, const int32_t* ia, const int32_t* ib, const double* meas,
                          const uint8_t* kind, int method, int world, int rank, int row_align, ShardStructure* S,
                          const std::vector<int32_t>* tile_breaks) {
  if (N <= 0 || E < 0 || world < 1 || rank < 0 || rank >= world || row_align < 1)
    return fail(PGO_ERR_INVALID_ARG, "build_shard_structure: bad sizes");
  for (int32_t e = 0; e < E; ++e) {
    if (ia[e] < 0 || ia[e] >= N || ib[e] < 0 || ib[e] >= N)
      return fail(PGO_ERR_INVALID_ARG, "edge " + std::to_string(e) + ": endpoint out of range");
    if (ia[e] == ib[e])
      return fail(PGO_ERR_INVALID_ARG, "edge " + std::to_string(e) + ": self loop (Ceres rejects duplicate parameter blocks)");
  }
  S->n_poses = N;
  S->world = world;
  S->rank = rank;
  S->rows_per_rank = pgo::rows_per_rank(N, world, row_align);
  S->lo = std::min<int64_t>((int64_t)rank * S->rows_per_rank, N);
  S->hi = std::min<int64_t>((int64_t)(rank + 1) * S->rows_per_rank, N);
  S->n_loc = S->hi - S->lo;
  const int32_t lo = S->lo, hi = S->hi;
  auto owned = [&](int32_t p) { return p >= lo && p < hi; };

  // local edges, sorted by (min endpoint, max endpoint, original index): pose
  // gathers of consecutive lanes then walk the pose array almost sequentially
  // (a counting sort on the smaller endpoint, then the handful of edges per bucket by (larger endpoint, index): the
  // comparison sort of the 4M edges of the 1M-pose graph was half of the symbolic phase)
  std::vector<int32_t> loc;
  {
    std::vector<int32_t> cnt((size_t)N + 1, 0);
    int64_t n_loc_e = 0;
    for (int32_t e = 0; e < E; ++e)
      if (owned(ia[e]) || owned(ib[e])) {
        cnt[(size_t)std::min(ia[e], ib[e]) + 1]++;
        ++n_loc_e;
      }
    for (int32_t i = 0; i < N; ++i) cnt[(size_t)i + 1] += cnt[i];
    loc.resize((size_t)n_loc_e);
    {
      std::vector<int32_t> fill(cnt.begin(), cnt.end() - 1);
      for (int32_t e = 0; e < E; ++e)   // ascending e: equal keys stay in index order
        if (owned(ia[e]) || owned(ib[e])) loc[(size_t)fill[std::min(ia[e], ib[e])]++] = e;
    }
    for (int32_t i = 0; i < N; ++i) {
      const int32_t b = cnt[i], en = cnt[(size_t)i + 1];
      if (en - b < 2) continue;
      std::sort(loc.begin() + b, loc.begin() + en, [&](int32_t x, int32_t y) {
        const int32_t xM = std::max(ia[x], ib[x]), yM = std::max(ia[y], ib[y]);
        if (xM != yM) return xM < yM;
        return x < y;
      });
    }
  }
  const int32_t EL = (int32_t)loc.size();
  S->n_edges_local = EL;
  S->orig_edge = loc;
  S->ia.resize(EL);
  S->ib.resize(EL);
  S->mx.resize(EL);
  S->my.resize(EL);
  S->mt.resize(EL);
  S->flags.resize(EL);
  S->n_cut = 0;
  for (int32_t k = 0; k < EL; ++k) {
    int32_t e = loc[k];
    S->ia[k] = ia[e];
    S->ib[k] = ib[e];
    S->mx[k] = meas[3 * (size_t)e + 0];
    S->my[k] = meas[3 * (size_t)e + 1];
    S->mt[k] = meas[3 * (size_t)e + 2];
    // DCS on closure + bogus edges only when METHOD == 1 (reference main.cpp:112-114,135-137)
    // ... and the switchable functor on the same edges when METHOD == 2 (main.cpp:115-125,138-145)
    uint8_t f = ((method == 1 || method == 2) && kind[e] != PGO_EDGE_ODOMETRY) ? 1 : 0;
    if (owned(ia[e])) f |= 2;
    S->flags[k] = f;
    if (!(owned(ia[e]) && owned(ib[e]))) S->n_cut++;
  }

  // incidences of owned rows
  std::vector<int32_t>& ptr = S->inc_ptr;
  ptr.assign((size_t)S->n_loc + 1, 0);
  for (int32_t k = 0; k < EL; ++k) {
    if (owned(S->ia[k])) ptr[S->ia[k] - lo + 1]++;
    if (owned(S->ib[k])) ptr[S->ib[k] - lo + 1]++;
  }
  for (int32_t i = 0; i < S->n_loc; ++i) ptr[i + 1] += ptr[i];
  S->n_inc = ptr[S->n_loc];
  S->n_inc_real = S->n_inc;
  S->padded = false;
  S->inc_edge.resize(S->n_inc);
  S->inc_col.resize(S->n_inc);
  {
    std::vector<int32_t> fill(ptr.begin(), ptr.end() - 1);
    // local edges are visited in sorted order, so each row's incidences come out
    // ordered by (min, max) endpoint, i.e. by column for the b-side and the a-side
    for (int32_t k = 0; k < EL; ++k) {
      int32_t a = S->ia[k], b = S->ib[k];
      if (owned(a)) {
        int32_t q = fill[a - lo]++;
        S->inc_edge[q] = (k << 1) | 0;
        S->inc_col[q] = b;
      }
      if (owned(b)) {
        int32_t q = fill[b - lo]++;
        S->inc_edge[q] = (k << 1) | 1;
        S->inc_col[q] = a;
      }
    }
    // order each row's incidences by column (deterministic accumulation order
    // that does not depend on the sharding)
    std::vector<std::pair<int32_t, int32_t>> tmp;   // one buffer for all rows
    for (int32_t i = 0; i < S->n_loc; ++i) {
      int32_t b = ptr[i], e = ptr[i + 1];
      if (e - b < 2) continue;
      tmp.resize((size_t)(e - b));
      for (int32_t q = b; q < e; ++q) tmp[q - b] = {S->inc_col[q], S->inc_edge[q]};
      std::sort(tmp.begin(), tmp.end(), [&](const std::pair<int32_t, int32_t>& x, const std::pair<int32_t, int32_t>& y) {
        if (x.first != y.first) return x.first < y.first;
        // duplicates of one pair: order by the caller's edge index
        return S->orig_edge[x.second >> 1] < S->orig_edge[y.second >> 1];
      });
      for (int32_t q = b; q < e; ++q) {
        S->inc_col[q] = tmp[q - b].first;
        S->inc_edge[q] = tmp[q - b].second;
      }
    }
  }

  build_halo_lists(E, ia, ib, S);

  // tiles
  S->tile_row.clear();
  S->tile_row.push_back(0);
  int32_t row = 0;
  size_t nb = 0;  // next tile break (global rows, ascending): a batched handle starts a tile at every problem start
  while (row < S->n_loc) {
    int32_t begin = row;
    int64_t inc0 = ptr[row];
    while (tile_breaks && nb < tile_breaks->size() && (*tile_breaks)[nb] <= lo + row) ++nb;
    const int32_t stop = (tile_breaks && nb < tile_breaks->size()) ? (*tile_breaks)[nb] - lo : S->n_loc;
    // at least one row per tile; more while their incidences fit one chunk
    ++row;
    while (row < S->n_loc && row < stop && ptr[row + 1] - inc0 <= TILE_INC && row - begin < TILE_INC) ++row;
    S->tile_row.push_back(row);
  }
  if (S->n_loc == 0) S->tile_row.assign(1, 0);
  // row of every incidence relative to its tile's first row (a tile has at most TILE_INC rows): K2's lanes read it
  // instead of searching inc_ptr
  S->inc_rowoff.assign((size_t)S->n_inc, 0);
  for (int32_t t = 0; t + 1 < (int32_t)S->tile_row.size(); ++t)
    for (int32_t r = S->tile_row[t]; r < S->tile_row[t + 1]; ++r)
      for (int32_t q = ptr[r]; q < ptr[r + 1]; ++q) S->inc_rowoff[q] = (uint8_t)(r - S->tile_row[t]);
  return PGO_OK;
}

// Every tile's incidences moved into a slot range of their own: tile t owns [TILE_INC t, TILE_INC (t + 1)), its real
// incidences first, NULL incidences behind them (edge -1, column = the tile's last row itself, a zero block that K2 never
// writes).  The nulls belong to the tile's last row, so inc_ptr stays a plain CSR row pointer and every loop over a row's
// incidences stays valid: a null adds 0 to whatever is summed.  What it buys: a workgroup knows where its tile's column indices
// and blocks are from its tile number alone -- K3 (k_spmv_1) requests them without waiting for the tile's descriptor, one
// dependent round trip less per tile: 140-146 -> 128-132 us at 1M poses (timing experiment before the layout existed), for
// 1.7 % more slots than incidences there.  Plain tiles only (<= TILE_INC incidences, <= TILE_INC / 3 rows); false = left as is.
bool pad_tiles_to_slots(ShardStructure* S) {
  const int32_t nt = S->n_tiles();
  if (S->padded || nt <= 0 || S->n_loc <= 0) return false;
  const std::vector<int32_t>& ptr = S->inc_ptr;
  for (int32_t t = 0; t < nt; ++t) {
    const int32_t r0 = S->tile_row[t], r1 = S->tile_row[t + 1];
    if (ptr[r1] - ptr[r0] > TILE_INC || (r1 - r0) * 3 > TILE_INC) return false;
  }
  if ((int64_t)nt * TILE_INC > (int64_t)INT32_MAX) return false;
  const int64_t slots = (int64_t)nt * TILE_INC;
  std::vector<int32_t> nptr((size_t)S->n_loc + 1), nedge((size_t)slots, -1), ncol((size_t)slots, 0);
  std::vector<uint8_t> noff((size_t)slots, 0);
  parallel_ranges(nt, [&](int64_t tb, int64_t te) {
    for (int64_t t = tb; t < te; ++t) {
      const int32_t r0 = S->tile_row[t], r1 = S->tile_row[t + 1], q0 = ptr[r0], nq = ptr[r1] - q0;
      const int32_t base = (int32_t)(t * TILE_INC);
      for (int32_t r = r0; r < r1; ++r) {
        nptr[r] = base + (ptr[r] - q0);
        for (int32_t q = ptr[r]; q < ptr[r + 1]; ++q) {
          nedge[base + (q - q0)] = S->inc_edge[q];
          ncol[base + (q - q0)] = S->inc_col[q];
          noff[base + (q - q0)] = (uint8_t)(r - r0);
        }
      }
      for (int32_t k = nq; k < TILE_INC; ++k) {
        ncol[base + k] = S->lo + (r1 - 1);
        noff[base + k] = (uint8_t)(r1 - 1 - r0);
      }
    }
  });
  nptr[S->n_loc] = (int32_t)slots;
  S->inc_ptr.swap(nptr);
  S->inc_edge.swap(nedge);
  S->inc_col.swap(ncol);
  S->inc_rowoff.swap(noff);
  S->n_inc = slots;
  S->padded = true;
  return true;
}

// Processing order of the row tiles for the SpMV (K3): breadth-first over the tile graph, so that tiles that run at
// the same time on one XCD (xcd_range deals each XCD one contiguous eighth of the order) gather the same lines of the
// search direction from that XCD's L2.  Memory layout, row order and every sum are untouched: a tile writes its own
// rows whatever its turn.  Tile graph: tile +- 1 (the odometry chain) and every SUPPORTED loop edge -- an edge (a, b)
// counts only if some other edge joins a' in {a-1, a, a+1} with b' in {b-1, b, b+1}; real revisits come in such
// runs, isolated false loops (the reference's add_random_C draws uniform pairs, g2o_util.h:151-171) do not, and they
// would turn the graph into a small world without any locality.
void compute_tile_order(const ShardStructure& S, std::vector<int32_t>* order) {
  const int32_t nt = S.n_tiles();
  order->clear();
  if (nt <= 0) return;
  order->reserve(nt);
  std::vector<int32_t> tile_of((size_t)S.n_loc);
  for (int32_t t = 0; t < nt; ++t)
    for (int32_t r = S.tile_row[t]; r < S.tile_row[t + 1]; ++r) tile_of[r] = t;
  const int32_t lo = S.lo, hi = S.hi;
  auto has_near = [&](int32_t row, int32_t col, int32_t skip_col) {  // row is local; any incidence with column in col +- 1?
    if (row < 0 || row >= S.n_loc) return false;
    for (int32_t q = S.inc_ptr[row]; q < S.inc_ptr[row + 1]; ++q) {
      const int32_t c = S.inc_col[q];
      if (c >= col - 1 && c <= col + 1 && c != skip_col) return true;
    }
    return false;
  };
  std::vector<char> seen(nt, 0);
  std::vector<int32_t> stamp(nt, -1);
  size_t head = 0;
  for (int32_t start = 0; start < nt; ++start) {
    if (seen[start]) continue;
    seen[start] = 1;
    order->push_back(start);
    while (head < order->size()) {
      const int32_t t = (*order)[head++];
      auto visit = [&](int32_t u) {
        if (u >= 0 && u < nt && !seen[u]) {
          seen[u] = 1;
          order->push_back(u);
        }
      };
      visit(t - 1);
      visit(t + 1);
      for (int32_t r = S.tile_row[t]; r < S.tile_row[t + 1]; ++r)
        for (int32_t q = S.inc_ptr[r]; q < S.inc_ptr[r + 1]; ++q) {
          const int32_t c = S.inc_col[q];
          if (c < lo || c >= hi) continue;
          const int32_t u = tile_of[c - lo];
          if (u == t || seen[u] || stamp[u] == t) continue;
          // supported by a parallel edge?  (r-1 | r | r+1) x (c-1 | c | c+1), not the edge itself
          const bool ok = has_near(r - 1, c, -1) || has_near(r + 1, c, -1) || has_near(r, c, c);
          if (ok) visit(u);
          else stamp[u] = t;  // do not test the same unsupported pair of tiles again for this tile
        }
    }
  }
}

}  // namespace pgo

extern "C" int pgo_shard_plan(int32_t n_poses, int32_t n_edges, const int32_t* ia, const int32_t* ib, int world,
                              int rank, int row_align, int32_t* lo, int32_t* hi, int32_t* n_local_edges,
                              int32_t* n_cut_edges) {
  if (n_poses <= 0 || n_edges < 0 || world < 1 || rank < 0 || rank >= world || row_align < 1 || (n_edges && (!ia || !ib)))
    return pgo::fail(PGO_ERR_INVALID_ARG, "pgo_shard_plan: bad argument");
  int32_t rpr = pgo::rows_per_rank(n_poses, world, row_align);
  int64_t l = std::min<int64_t>((int64_t)rank * rpr, n_poses), h = std::min<int64_t>((int64_t)(rank + 1) * rpr, n_poses);
  int32_t nl = 0, nc = 0;
  for (int32_t e = 0; e < n_edges; ++e) {
    bool oa = ia[e] >= l && ia[e] < h, ob = ib[e] >= l && ib[e] < h;
    if (oa || ob) {
      ++nl;
      if (!(oa && ob)) ++nc;
    }
  }
  if (lo) *lo = (int32_t)l;
  if (hi) *hi = (int32_t)h;
  if (n_local_edges) *n_local_edges = nl;
  if (n_cut_edges) *n_cut_edges = nc;
  return PGO_OK;
}

extern "C" int pgo_shard_halo(int32_t n_poses, int32_t n_edges, const int32_t* ia, const int32_t* ib, int world, int rank,
                              int row_align, int64_t* send_rows, int64_t* recv_rows) {
  if (n_poses <= 0 || n_edges < 0 || world < 1 || rank < 0 || rank >= world || row_align < 1 || (n_edges && (!ia || !ib)) ||
      !send_rows || !recv_rows)
    return pgo::fail(PGO_ERR_INVALID_ARG, "pgo_shard_halo: bad argument");
  for (int32_t e = 0; e < n_edges; ++e)
    if (ia[e] < 0 || ia[e] >= n_poses || ib[e] < 0 || ib[e] >= n_poses)
      return pgo::fail(PGO_ERR_INVALID_ARG, "pgo_shard_halo: endpoint out of range");
  pgo::ShardStructure S;
  S.n_poses = n_poses;
  S.world = world;
  S.rank = rank;
  S.rows_per_rank = pgo::rows_per_rank(n_poses, world, row_align);
  pgo::build_halo_lists(n_edges, ia, ib, &S);
  for (int s = 0; s < world; ++s) {
    send_rows[s] = S.halo_send_off[s + 1] - S.halo_send_off[s];
    recv_rows[s] = S.halo_recv_off[s + 1] - S.halo_recv_off[s];
  }
  return PGO_OK;
}

extern "C" int pgo_pose_order(int32_t n_poses, int32_t n_edges, const int32_t* ia, const int32_t* ib, int32_t segment,
                              int32_t* perm) {
  if (n_poses <= 0 || n_edges < 0 || segment < 1 || !perm || (n_edges && (!ia || !ib)))
    return pgo::fail(PGO_ERR_INVALID_ARG, "pgo_pose_order: bad argument");
  for (int32_t e = 0; e < n_edges; ++e)
    if (ia[e] < 0 || ia[e] >= n_poses || ib[e] < 0 || ib[e] >= n_poses)
      return pgo::fail(PGO_ERR_INVALID_ARG, "pgo_pose_order: endpoint out of range");
  std::vector<int32_t> p;
  int st = pgo::compute_pose_order(n_poses, n_edges, ia, ib, segment, &p);
  if (st != PGO_OK) return st;
  std::copy(p.begin(), p.end(), perm);
  return PGO_OK;
}
