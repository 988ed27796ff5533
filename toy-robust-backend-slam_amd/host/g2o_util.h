// ReadG2O with the reference's public surface (DCS-ceres/include/g2o_util.h:20-188), implemented on
// the C-ABI of libpgo.so (no Boost).  Differences, all deliberate: I/O and parse errors throw
// std::runtime_error instead of being ignored; add_random_C takes an optional seed (reference main.cpp:43
// seeds with time(0) once per run); the writers take an optional precision.
#ifndef PGO_HOST_G2O_UTIL_H_
#define PGO_HOST_G2O_UTIL_H_

#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "graph.h"
#include "pgo.h"

#define ODOMETRY_EDGE 0
#define CLOSURE_EDGE 1
#define BOGUS_EDGE 2

class ReadG2O {
 public:
  explicit ReadG2O(const std::string& fName) {
    check(pgo_g2o_load(fName.c_str(), &g_));
    const int n = pgo_graph_num_poses(g_);
    const int32_t* ids = pgo_graph_pose_ids(g_);
    const double* p = pgo_graph_poses(g_);
    for (int i = 0; i < n; ++i) nNodes.push_back(new Node(ids[i], p[3 * i], p[3 * i + 1], p[3 * i + 2]));
    rebuild_edges(0);
  }
  ~ReadG2O() {
    for (Node* n : nNodes) delete n;
    for (auto* v : {&nEdgesOdometry, &nEdgesClosure, &nEdgesBogus})
      for (Edge* e : *v) delete e;
    pgo_graph_free(g_);
  }
  ReadG2O(const ReadG2O&) = delete;
  ReadG2O& operator=(const ReadG2O&) = delete;

  // g2o_util.h:151-171.  seed < 0: srand(time(0)) as the reference's main does.
  void add_random_C(int count, long long seed = -1) {
    std::cout << "Adding Bogus edges as described in Vertigo paper" << std::endl;
    const int before = pgo_graph_num_edges(g_);
    check(pgo_inject_outliers(g_, count, seed));
    rebuild_edges(before);
    for (size_t i = nEdgesBogus.size() - (size_t)count; i < nEdgesBogus.size(); ++i)
      std::cout << "  " << nEdgesBogus[i]->a->index << "<--->" << nEdgesBogus[i]->b->index << std::endl;
  }

  // g2o_util.h:93-112: "<index> <x> <y> <theta>" / "<a> <b> <type>"
  void writePoseGraph_nodes(const std::string& fname, int precision = 0) {
    std::cout << "writePoseGraph nodes: " << fname << std::endl;
    sync_to_graph();
    check(pgo_write_nodes(g_, fname.c_str(), precision));
  }
  void writePoseGraph_edges(const std::string& fname) {
    std::cout << "writePoseGraph Edges : " << fname << std::endl;
    check(pgo_write_edges(g_, fname.c_str()));
  }

  // g2o_util.h:114-148 of the reference: priors[i] / *optimized[i] per closure then bogus edge
  void writePoseGraph_switches(const std::string& fname, std::vector<double>& priors, std::vector<double*>& optimized) {
    std::cout << "#Closure Edges : " << nEdgesClosure.size() << std::endl;
    std::cout << "#Bogus Edges : " << nEdgesBogus.size() << std::endl;
    std::cout << "#priors : " << priors.size() << std::endl;
    std::cout << "#optimized " << optimized.size() << std::endl;
    std::vector<double> sw(nEdgesOdometry.size(), 1.0);
    for (double* s : optimized) sw.push_back(*s);
    check(pgo_write_switches(g_, fname.c_str(), sw.data()));
  }

  std::vector<Node*> nNodes;
  std::vector<Edge*> nEdgesOdometry;
  std::vector<Edge*> nEdgesClosure;
  std::vector<Edge*> nEdgesBogus;

  pgo_graph* handle() { sync_to_graph(); return g_; }

 private:
  static void check(int st) {
    if (st != PGO_OK) throw std::runtime_error(std::string("pgo: ") + pgo_strerror(st) + ": " + pgo_last_error());
  }
  void sync_to_graph() {  // Node::p is the live storage; mirror it into the flat array
    double* p = pgo_graph_poses(g_);
    for (size_t i = 0; i < nNodes.size(); ++i)
      for (int k = 0; k < 3; ++k) p[3 * i + k] = nNodes[i]->p[k];
  }
  void rebuild_edges(int /*first_new*/) {
    for (auto* v : {&nEdgesOdometry, &nEdgesClosure, &nEdgesBogus}) {
      for (Edge* e : *v) delete e;
      v->clear();
    }
    const int E = pgo_graph_num_edges(g_);
    const int32_t *ia = pgo_graph_edge_a(g_), *ib = pgo_graph_edge_b(g_);
    const double *m = pgo_graph_edge_meas(g_), *q = pgo_graph_edge_info(g_);
    const uint8_t* kind = pgo_graph_edge_kind(g_);
    for (int e = 0; e < E; ++e) {
      Edge* ed = new Edge(nNodes[ia[e]], nNodes[ib[e]], kind[e]);
      ed->setEdgePose(m[3 * e], m[3 * e + 1], m[3 * e + 2]);
      ed->setInformationMatrix(q[6 * e], q[6 * e + 1], q[6 * e + 2], q[6 * e + 3], q[6 * e + 4], q[6 * e + 5]);
      (kind[e] == ODOMETRY_EDGE ? nEdgesOdometry : kind[e] == CLOSURE_EDGE ? nEdgesClosure : nEdgesBogus).push_back(ed);
    }
  }
  pgo_graph* g_ = nullptr;
};

#endif
