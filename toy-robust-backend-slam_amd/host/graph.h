// Host-side pose-graph types with the reference's names and members
// (reference: DCS-ceres/include/graph.h:4-56), so code written against the reference's
// Node / Edge compiles against this backend.  Unlike the reference, Node owns and frees p.
#ifndef PGO_HOST_GRAPH_H_
#define PGO_HOST_GRAPH_H_

struct Node {
  int index;   // id as read from the g2o file
  double* p;   // (x, y, theta): the parameter block the solver updates in place

  Node(int index_, double x, double y, double theta) : index(index_), p(new double[3]{x, y, theta}) {}
  ~Node() { delete[] p; }
  Node(const Node&) = delete;
  Node& operator=(const Node&) = delete;
};

struct Edge {
  const Node* a;
  const Node* b;
  double x = 0, y = 0, theta = 0;                           // measured pose of b in the frame of a
  double I11 = 0, I12 = 0, I13 = 0, I22 = 0, I23 = 0, I33 = 0;  // information matrix: parsed, unused by METHOD 0/1
  int edge_type;                                            // 0 odometry, 1 loop closure, 2 bogus

  Edge(const Node* a_, const Node* b_, int type) : a(a_), b(b_), edge_type(type) {}
  void setEdgePose(double x_, double y_, double theta_) { x = x_; y = y_; theta = theta_; }
  void setInformationMatrix(double i11, double i12, double i13, double i22, double i23, double i33) {
    I11 = i11; I12 = i12; I13 = i13; I22 = i22; I23 = i23; I33 = i33;
  }
};

#endif
