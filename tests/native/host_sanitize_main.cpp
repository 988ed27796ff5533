// Host-only exerciser for the AddressSanitizer / UBSan build of the graph + shard-structure code
// (tests/test_host_sanitizers.py compiles host_graph.cpp + structure.cpp + this file with g++ -fsanitize=...;
// GPU sanitizers are not available on the pool, so the device code is covered by the parity tests instead).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "pgo_internal.h"

#define CHECK(c)                                                    \
  do {                                                              \
    if (!(c)) {                                                     \
      fprintf(stderr, "CHECK failed: %s (line %d)\n", #c, __LINE__); \
      return 1;                                                     \
    }                                                               \
  } while (0)

int main(int argc, char** argv) {
  const char* data_dir = argc > 1 ? argv[1] : ".";
  pgo_graph* g = nullptr;
  CHECK(pgo_g2o_load((std::string(data_dir) + "/INTEL.g2o").c_str(), &g) == PGO_OK);
  CHECK(pgo_graph_num_poses(g) == 1228 && pgo_graph_num_edges(g) == 1483);
  CHECK(pgo_inject_outliers(g, 50, 1) == PGO_OK);
  CHECK(pgo_graph_num_edges_of_kind(g, PGO_EDGE_BOGUS) == 50);
  CHECK(pgo_write_nodes(g, "/tmp/pgo_san_nodes.txt", 0) == PGO_OK);
  CHECK(pgo_write_edges(g, "/tmp/pgo_san_edges.txt") == PGO_OK);
  CHECK(pgo_write_g2o(g, "/tmp/pgo_san.g2o") == PGO_OK);
  pgo_graph* back = nullptr;
  CHECK(pgo_g2o_load("/tmp/pgo_san.g2o", &back) == PGO_OK);
  CHECK(pgo_graph_num_edges(back) == pgo_graph_num_edges(g));
  pgo_graph_free(back);
  // malformed inputs
  pgo_graph* bad = nullptr;
  const char* t1 = "VERTEX_SE2 0 0 0 0\nEDGE_SE2 0 9 1 0 0 1 0 0 1 0 1\n";
  CHECK(pgo_g2o_parse(t1, strlen(t1), &bad) == PGO_ERR_PARSE && bad == nullptr);
  const char* t2 = "VERTEX_SE2 0 0 0\n";
  CHECK(pgo_g2o_parse(t2, strlen(t2), &bad) == PGO_ERR_PARSE);
  CHECK(pgo_g2o_parse("", 0, &bad) == PGO_OK && pgo_graph_num_poses(bad) == 0);
  pgo_graph_free(bad);
  CHECK(pgo_g2o_load("/nonexistent/x.g2o", &bad) == PGO_ERR_IO);

  // shard structures of the INTEL graph and of a synthetic one, several world sizes / alignments
  for (int pass = 0; pass < 2; ++pass) {
    pgo_graph* h = g;
    if (pass == 1) CHECK(pgo_synth_manhattan(30011, 4.0, 0.1, 7, &h) == PGO_OK);
    const pgo::Graph& G = h->g;
    for (int world : {1, 2, 3, 8})
      for (int align : {1, 4, 32}) {
        int64_t rows = 0, cost_edges = 0, inc = 0;
        std::vector<int64_t> snd_total(world, 0), rcv_total(world, 0);
        for (int rank = 0; rank < world; ++rank) {
          pgo::ShardStructure S;
          CHECK(pgo::build_shard_structure(G.n_poses(), G.n_edges(), G.ea.data(), G.eb.data(), G.meas.data(), G.kind.data(), 1,
                                           world, rank, align, &S) == PGO_OK);
          rows += S.n_loc;
          inc += S.n_inc;
          for (uint8_t f : S.flags) cost_edges += (f >> 1) & 1;
          CHECK((int64_t)S.inc_ptr.back() == S.n_inc && S.tile_row.back() == S.n_loc);
          for (int t = 0; t < S.n_tiles(); ++t) {
            const int r0 = S.tile_row[t], r1 = S.tile_row[t + 1];
            CHECK(r1 > r0 && (S.inc_ptr[r1] - S.inc_ptr[r0] <= pgo::TILE_INC || r1 - r0 == 1));
          }
          {  // the padded-slot layout of large graphs (pad_tiles_to_slots), on a copy: same incidences, nulls behind them
            pgo::ShardStructure P = S;
            const bool plain = pgo::pad_tiles_to_slots(&P);
            if (plain) {
              CHECK(P.padded && P.n_inc == (int64_t)P.n_tiles() * pgo::TILE_INC && P.n_inc_real == S.n_inc);
              CHECK((int64_t)P.inc_ptr.back() == P.n_inc && (int64_t)P.inc_edge.size() == P.n_inc && (int64_t)P.inc_rowoff.size() == P.n_inc);
              int64_t real = 0;
              for (int t = 0; t < P.n_tiles(); ++t) {
                const int r0 = P.tile_row[t], r1 = P.tile_row[t + 1];
                CHECK(P.inc_ptr[r0] == t * pgo::TILE_INC);
                for (int r = r0; r < r1; ++r) {
                  const int nreal = S.inc_ptr[r + 1] - S.inc_ptr[r];
                  for (int k = 0; k < nreal; ++k) {   // the row's real incidences, in order, at the start of its slot range
                    CHECK(P.inc_edge[P.inc_ptr[r] + k] == S.inc_edge[S.inc_ptr[r] + k] && P.inc_col[P.inc_ptr[r] + k] == S.inc_col[S.inc_ptr[r] + k]);
                    CHECK(P.inc_rowoff[P.inc_ptr[r] + k] == r - r0);
                  }
                  real += nreal;
                  for (int q = P.inc_ptr[r] + nreal; q < P.inc_ptr[r + 1]; ++q)   // nulls: only behind a tile's last row
                    CHECK(r == r1 - 1 && P.inc_edge[q] == -1 && P.inc_col[q] == P.lo + r && P.inc_rowoff[q] == r - r0);
                }
              }
              CHECK(real == S.n_inc);
              CHECK(!pgo::pad_tiles_to_slots(&P));   // idempotent: a padded structure is left alone
            } else {
              CHECK(!P.padded && P.n_inc == S.n_inc);
            }
          }
          for (int s = 0; s < world; ++s) {
            snd_total[rank] += S.halo_send_off[s + 1] - S.halo_send_off[s];
            rcv_total[rank] += S.halo_recv_off[s + 1] - S.halo_recv_off[s];
          }
        }
        CHECK(rows == G.n_poses() && cost_edges == G.n_edges() && inc == 2 * (int64_t)G.n_edges());
        int64_t a = 0, b = 0;
        for (int r = 0; r < world; ++r) { a += snd_total[r]; b += rcv_total[r]; }
        CHECK(a == b);
      }
    if (pass == 1) pgo_graph_free(h);
  }
  // invalid structure requests
  {
    pgo::ShardStructure S;
    int32_t ia[1] = {0}, ib[1] = {0};
    double m[3] = {0, 0, 0};
    uint8_t k[1] = {1};
    CHECK(pgo::build_shard_structure(2, 1, ia, ib, m, k, 1, 1, 0, 1, &S) == PGO_ERR_INVALID_ARG);  // self loop
    ib[0] = 5;
    CHECK(pgo::build_shard_structure(2, 1, ia, ib, m, k, 1, 1, 0, 1, &S) == PGO_ERR_INVALID_ARG);  // out of range
  }
  pgo_graph_free(g);
  printf("host sanitizer run ok\n");
  return 0;
}
