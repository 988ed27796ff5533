// Host-side graph I/O: g2o loader + classifier, outlier injector, text writers and
// the synthetic Manhattan-world generator.  Pure C++17, no GPU, no Boost.
//
// Behavioural source (all relative to /root/reference/DCS-ceres):
//   loader / classifier   include/g2o_util.h:23-89
//   injector              include/g2o_util.h:151-171
//   writers               include/g2o_util.h:93-112,179-186
//   residual-block order  main.cpp:95-150
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "pgo_internal.h"

namespace pgo {

static thread_local std::string g_last_error;
int fail(int status, const std::string& msg) {
  g_last_error = msg;
  return status;
}
const std::string& last_error() { return g_last_error; }

void Graph::insert_edge(int32_t a, int32_t b, const double m[3], const double inf[6], int k) {
  // position right after the last edge of the same kind group
  size_t pos = 0;
  for (int j = 0; j <= k; ++j) pos += (size_t)n_kind[j];
  ea.insert(ea.begin() + pos, a);
  eb.insert(eb.begin() + pos, b);
  meas.insert(meas.begin() + 3 * pos, m, m + 3);
  info.insert(info.begin() + 6 * pos, inf, inf + 6);
  kind.insert(kind.begin() + pos, (uint8_t)k);
  n_kind[k]++;
}

// ------------------------------------------------------------------ parsing
namespace {

struct Cursor {
  const char* p;
  const char* end;
};

inline void skip_spaces(Cursor& c) {
  while (c.p < c.end && *c.p == ' ') ++c.p;
}
// token = maximal run of non-space characters (the reference splits on ' ' with
// token compression, g2o_util.h:36)
inline bool next_token(Cursor& c, const char** tb, const char** te) {
  skip_spaces(c);
  if (c.p >= c.end) return false;
  *tb = c.p;
  while (c.p < c.end && *c.p != ' ') ++c.p;
  *te = c.p;
  return true;
}
inline bool tok_eq(const char* tb, const char* te, const char* s) {
  size_t n = strlen(s);
  return (size_t)(te - tb) == n && memcmp(tb, s, n) == 0;
}
// lexical_cast semantics: the WHOLE token must convert
inline bool to_int(const char* tb, const char* te, int32_t* out) {
  char buf[64];
  size_t n = (size_t)(te - tb);
  if (n == 0 || n >= sizeof buf) return false;
  memcpy(buf, tb, n);
  buf[n] = 0;
  char* endp = nullptr;
  errno = 0;
  long v = strtol(buf, &endp, 10);
  if (endp != buf + n || errno != 0 || v < INT32_MIN || v > INT32_MAX) return false;
  *out = (int32_t)v;
  return true;
}
inline bool to_double(const char* tb, const char* te, double* out) {
  char buf[128];
  size_t n = (size_t)(te - tb);
  if (n == 0 || n >= sizeof buf) return false;
  memcpy(buf, tb, n);
  buf[n] = 0;
  char* endp = nullptr;
  double v = strtod(buf, &endp);
  if (endp != buf + n) return false;
  *out = v;
  return true;
}

}  // namespace

static int parse_g2o(const char* text, size_t len, Graph* g) {
  // Two groups are collected separately and concatenated at the end so that the
  // edge order is odometry..., closure... (each in file order), as the
  // reference's two vectors are (g2o_util.h:73,80).
  std::vector<int32_t> ea[2], eb[2];
  std::vector<double> meas[2], info[2];
  const char* p = text;
  const char* end = text + len;
  int64_t lineno = 0;
  while (p < end) {
    const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
    const char* le = nl ? nl : end;
    ++lineno;
    Cursor c{p, le};
    // tolerate CRLF files (the reference would throw bad_lexical_cast on them)
    if (c.end > c.p && c.end[-1] == '\r') --c.end;
    const char *tb, *te;
    // NB boost::split keeps an empty leading token when the line starts with a
    // space, so such a line matches no tag in the reference; mirror that.
    bool leading_space = (c.p < c.end && *c.p == ' ');
    if (!leading_space && next_token(c, &tb, &te)) {
      bool is_v = tok_eq(tb, te, "VERTEX_SE2") || tok_eq(tb, te, "VERTEX2");
      bool is_e = !is_v && (tok_eq(tb, te, "EDGE_SE2") || tok_eq(tb, te, "EDGE2"));
      if (is_v) {
        int32_t id;
        double v[3];
        const char *a, *b;
        if (!next_token(c, &a, &b) || !to_int(a, b, &id))
          return fail(PGO_ERR_PARSE, "g2o line " + std::to_string(lineno) + ": bad vertex id");
        for (int k = 0; k < 3; ++k)
          if (!next_token(c, &a, &b) || !to_double(a, b, &v[k]))
            return fail(PGO_ERR_PARSE, "g2o line " + std::to_string(lineno) + ": bad vertex field");
        g->pose_id.push_back(id);
        g->pose.insert(g->pose.end(), v, v + 3);
      } else if (is_e) {
        int32_t ab[2];
        double f[9];
        const char *a, *b;
        for (int k = 0; k < 2; ++k)
          if (!next_token(c, &a, &b) || !to_int(a, b, &ab[k]))
            return fail(PGO_ERR_PARSE, "g2o line " + std::to_string(lineno) + ": bad edge endpoint");
        for (int k = 0; k < 9; ++k)
          if (!next_token(c, &a, &b) || !to_double(a, b, &f[k]))
            return fail(PGO_ERR_PARSE, "g2o line " + std::to_string(lineno) + ": bad edge field");
        int32_t n = g->n_poses();
        // the reference indexes nNodes[] by the raw value (g2o_util.h:70,77): out of
        // range is undefined behaviour there, an error here
        if (ab[0] < 0 || ab[0] >= n || ab[1] < 0 || ab[1] >= n)
          return fail(PGO_ERR_PARSE, "g2o line " + std::to_string(lineno) +
                                         ": edge endpoint outside the vertices read so far");
        int grp = (std::abs(ab[0] - ab[1]) < 5) ? 0 : 1;  // g2o_util.h:68
        ea[grp].push_back(ab[0]);
        eb[grp].push_back(ab[1]);
        meas[grp].insert(meas[grp].end(), f, f + 3);
        info[grp].insert(info[grp].end(), f + 3, f + 9);
      }
    }
    p = nl ? nl + 1 : end;
  }
  for (int grp = 0; grp < 2; ++grp) {
    g->ea.insert(g->ea.end(), ea[grp].begin(), ea[grp].end());
    g->eb.insert(g->eb.end(), eb[grp].begin(), eb[grp].end());
    g->meas.insert(g->meas.end(), meas[grp].begin(), meas[grp].end());
    g->info.insert(g->info.end(), info[grp].begin(), info[grp].end());
    g->kind.insert(g->kind.end(), ea[grp].size(), (uint8_t)grp);
    g->n_kind[grp] = (int32_t)ea[grp].size();
  }
  g->n_kind[2] = 0;
  return PGO_OK;
}

// -------------------------------------------------------------- splitmix64
struct SplitMix64 {
  uint64_t s;
  explicit SplitMix64(uint64_t seed) : s(seed) {}
  uint64_t next() {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  double uniform() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }  // [0,1)
  double normal() {  // Box-Muller, one value per call (second value discarded: simple + reproducible)
    double u1 = uniform(), u2 = uniform();
    if (u1 < 1e-300) u1 = 1e-300;
    return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586476925 * u2);
  }
  uint64_t below(uint64_t n) { return next() % n; }
};

static inline double wrap_pi(double a) {
  while (a > M_PI) a -= 2.0 * M_PI;
  while (a <= -M_PI) a += 2.0 * M_PI;
  return a;
}

static int synth_manhattan(int32_t N, double edges_per_pose, double outlier_frac, uint64_t seed, Graph* g) {
  if (N < 2) return fail(PGO_ERR_INVALID_ARG, "synth: need at least 2 poses");
  SplitMix64 rng(seed);
  const double sig_xy = 0.02, sig_t = 0.01;
  // ground truth on the integer grid; heading in quarter turns
  std::vector<int32_t> gx(N), gy(N), gh(N);
  static const int dxs[4] = {1, 0, -1, 0}, dys[4] = {0, 1, 0, -1};
  // The world is a bounded square (a finite building): side ~ sqrt(N / 4), so that a cell is
  // revisited about four times and every pose finds earlier poses next to it.
  const int32_t side = std::max<int32_t>(8, (int32_t)std::ceil(std::sqrt((double)N / 4.0)));
  auto inside = [&](int32_t x, int32_t y) { return x >= 0 && x < side && y >= 0 && y < side; };
  gx[0] = side / 2;
  gy[0] = side / 2;
  gh[0] = 0;
  for (int32_t i = 0; i + 1 < N; ++i) {
    // move one unit along the current heading, then turn +-90 deg with p = 0.2; a heading
    // that would leave the square next step is turned away from the wall
    gx[i + 1] = gx[i] + dxs[gh[i]];
    gy[i + 1] = gy[i] + dys[gh[i]];
    int h = gh[i];
    double u = rng.uniform();
    if (u < 0.1) h = (h + 1) & 3;
    else if (u < 0.2) h = (h + 3) & 3;
    if (!inside(gx[i + 1] + dxs[h], gy[i + 1] + dys[h])) {
      const int first = (rng.next() & 1) ? 1 : 3;
      const int tries[3] = {(h + first) & 3, (h + 4 - first) & 3, (h + 2) & 3};
      for (int t = 0; t < 3; ++t)
        if (inside(gx[i + 1] + dxs[tries[t]], gy[i + 1] + dys[tries[t]])) {
          h = tries[t];
          break;
        }
    }
    gh[i + 1] = h;
  }
  auto true_rel = [&](int32_t a, int32_t b, double out[3]) {
    double dx = (double)(gx[b] - gx[a]), dy = (double)(gy[b] - gy[a]);
    int ha = gh[a];
    // rotate the world offset into frame a (heading is a multiple of 90 deg: exact)
    double c = (double)dxs[ha], s = (double)dys[ha];
    out[0] = c * dx + s * dy;
    out[1] = -s * dx + c * dy;
    out[2] = wrap_pi((double)(((gh[b] - ha) & 3)) * (M_PI / 2.0));
  };
  const double unit_info[6] = {1, 0, 0, 1, 0, 1};

  g->pose_id.resize(N);
  g->pose.assign((size_t)3 * N, 0.0);
  for (int32_t i = 0; i < N; ++i) g->pose_id[i] = i;

  // odometry edges (i, i+1) + dead-reckoned initial poses
  std::vector<int32_t> oa, ob;
  std::vector<double> om;
  oa.reserve(N);
  ob.reserve(N);
  om.reserve((size_t)3 * N);
  for (int32_t i = 0; i + 1 < N; ++i) {
    double m[3];
    true_rel(i, i + 1, m);
    m[0] += sig_xy * rng.normal();
    m[1] += sig_xy * rng.normal();
    m[2] += sig_t * rng.normal();
    oa.push_back(i);
    ob.push_back(i + 1);
    om.insert(om.end(), m, m + 3);
    const double* p = &g->pose[(size_t)3 * i];
    double c = std::cos(p[2]), s = std::sin(p[2]);
    double* q = &g->pose[(size_t)3 * (i + 1)];
    q[0] = p[0] + c * m[0] - s * m[1];
    q[1] = p[1] + s * m[0] + c * m[1];
    q[2] = p[2] + m[2];
  }

  // loop-closure candidates through a spatial hash of grid cells.  Each cell
  // remembers its first and its latest visitor; pose i looks at the 3x3 cells
  // around it (all within sqrt(2) < 1.5 m), keeps visitors j with i - j >= 5,
  // and draws up to 3 of them.
  struct Cell {
    int32_t first, last;
  };
  std::unordered_map<uint64_t, Cell> cells;
  cells.reserve((size_t)N);
  auto key = [](int32_t x, int32_t y) { return ((uint64_t)(uint32_t)x << 32) | (uint64_t)(uint32_t)y; };
  std::vector<int32_t> ca, cb;
  ca.reserve((size_t)3 * N);
  cb.reserve((size_t)3 * N);
  for (int32_t i = 0; i < N; ++i) {
    int32_t cand[18];
    int nc = 0;
    for (int dy = -1; dy <= 1; ++dy)
      for (int dx = -1; dx <= 1; ++dx) {
        auto it = cells.find(key(gx[i] + dx, gy[i] + dy));
        if (it == cells.end()) continue;
        int32_t two[2] = {it->second.first, it->second.last};
        for (int t = 0; t < 2; ++t) {
          int32_t j = two[t];
          if (i - j < 5) continue;
          bool dup = false;
          for (int q = 0; q < nc; ++q) dup |= (cand[q] == j);
          if (!dup) cand[nc++] = j;
        }
      }
    int take = nc < 3 ? nc : 3;
    for (int t = 0; t < take; ++t) {  // partial Fisher-Yates
      int r = t + (int)rng.below((uint64_t)(nc - t));
      int32_t tmp = cand[t];
      cand[t] = cand[r];
      cand[r] = tmp;
      ca.push_back(cand[t]);
      cb.push_back(i);
    }
    auto ins = cells.emplace(key(gx[i], gy[i]), Cell{i, i});
    if (!ins.second) ins.first->second.last = i;
  }
  // thin the candidates to the requested edge budget
  double want = (edges_per_pose - 1.0) * (double)N;
  if (want < 0) want = 0;
  double keep = ca.empty() ? 0.0 : want / (double)ca.size();
  std::vector<int32_t> la, lb;
  std::vector<double> lm;
  for (size_t k = 0; k < ca.size(); ++k) {
    if (keep < 1.0 && rng.uniform() >= keep) continue;
    double m[3];
    true_rel(ca[k], cb[k], m);
    m[0] += sig_xy * rng.normal();
    m[1] += sig_xy * rng.normal();
    m[2] = wrap_pi(m[2] + sig_t * rng.normal());
    la.push_back(ca[k]);
    lb.push_back(cb[k]);
    lm.insert(lm.end(), m, m + 3);
  }
  // assemble: odometry group first, then the closures (kind by the reference's
  // abs(a-b) < 5 rule, which they all fail by construction), then bogus loops
  g->ea = oa;
  g->eb = ob;
  g->meas = om;
  g->kind.assign(oa.size(), 0);
  g->n_kind[0] = (int32_t)oa.size();
  g->ea.insert(g->ea.end(), la.begin(), la.end());
  g->eb.insert(g->eb.end(), lb.begin(), lb.end());
  g->meas.insert(g->meas.end(), lm.begin(), lm.end());
  g->kind.insert(g->kind.end(), la.size(), 1);
  g->n_kind[1] = (int32_t)la.size();
  int64_t n_bogus = (int64_t)std::llround(outlier_frac * (double)la.size());
  for (int64_t k = 0; k < n_bogus; ++k) {
    int32_t a = (int32_t)rng.below((uint64_t)N);
    int32_t b = (int32_t)rng.below((uint64_t)N);
    if (a == b) b = (b + 1) % N;  // g2o_util.h:160-163
    g->ea.push_back(a);
    g->eb.push_back(b);
    g->meas.insert(g->meas.end(), 3, 0.0);  // g2o_util.h:166: integer division => 0
    g->kind.push_back(2);
  }
  g->n_kind[2] = (int32_t)n_bogus;
  g->info.resize((size_t)6 * g->ea.size());
  for (size_t e = 0; e < g->ea.size(); ++e) memcpy(&g->info[6 * e], unit_info, sizeof unit_info);
  return PGO_OK;
}

}  // namespace pgo

// ===================================================================== C-ABI
using pgo::fail;
using pgo::Graph;

extern "C" {

const char* pgo_strerror(int status) {
  switch (status) {
    case PGO_OK: return "ok";
    case PGO_ERR_INVALID_ARG: return "invalid argument";
    case PGO_ERR_IO: return "I/O error";
    case PGO_ERR_PARSE: return "g2o parse error";
    case PGO_ERR_NO_DEVICE: return "no gfx950 device";
    case PGO_ERR_HIP: return "HIP runtime error";
    case PGO_ERR_COMM: return "communicator error";
    case PGO_ERR_NUMERIC: return "non-finite residual or Jacobian";
    case PGO_ERR_UNSUPPORTED: return "unsupported";
    case PGO_ERR_NOMEM: return "out of memory";
    default: return "unknown status";
  }
}
const char* pgo_last_error(void) {
  return pgo::last_error().c_str();
}
const char* pgo_version(void) { return "pgo-amd 0.1 (gfx950)"; }

int pgo_g2o_parse(const char* text, size_t len, pgo_graph** out) {
  if (!text || !out) return fail(PGO_ERR_INVALID_ARG, "pgo_g2o_parse: null argument");
  std::unique_ptr<pgo_graph> pg(new pgo_graph);
  int st = pgo::parse_g2o(text, len, &pg->g);
  if (st != PGO_OK) return st;
  *out = pg.release();
  return PGO_OK;
}

int pgo_g2o_load(const char* path, pgo_graph** out) {
  if (!path || !out) return fail(PGO_ERR_INVALID_ARG, "pgo_g2o_load: null argument");
  FILE* f = fopen(path, "rb");
  if (!f) return fail(PGO_ERR_IO, std::string("cannot open ") + path + ": " + strerror(errno));
  std::string buf;
  char chunk[1 << 16];
  size_t n;
  while ((n = fread(chunk, 1, sizeof chunk, f)) > 0) buf.append(chunk, n);
  bool bad = ferror(f);
  fclose(f);
  if (bad) return fail(PGO_ERR_IO, std::string("read error on ") + path);
  return pgo_g2o_parse(buf.data(), buf.size(), out);
}

int pgo_graph_from_arrays(int32_t n_poses, const double* poses, int32_t n_edges, const int32_t* ia,
                          const int32_t* ib, const double* meas, const double* info6, const uint8_t* kind,
                          pgo_graph** out) {
  if (n_poses < 0 || n_edges < 0 || !out || (n_poses && !poses) || (n_edges && (!ia || !ib || !meas || !kind)))
    return fail(PGO_ERR_INVALID_ARG, "pgo_graph_from_arrays: null argument");
  std::unique_ptr<pgo_graph> pg(new pgo_graph);
  Graph& g = pg->g;
  g.pose_id.resize(n_poses);
  for (int32_t i = 0; i < n_poses; ++i) g.pose_id[i] = i;
  g.pose.assign(poses, poses + (size_t)3 * n_poses);
  const double unit_info[6] = {1, 0, 0, 1, 0, 1};
  for (int k = 0; k < 3; ++k)
    for (int32_t e = 0; e < n_edges; ++e) {
      if (kind[e] > 2) return fail(PGO_ERR_INVALID_ARG, "edge kind must be 0, 1 or 2");
      if (kind[e] != k) continue;
      if (ia[e] < 0 || ia[e] >= n_poses || ib[e] < 0 || ib[e] >= n_poses)
        return fail(PGO_ERR_INVALID_ARG, "edge endpoint out of range");
      g.ea.push_back(ia[e]);
      g.eb.push_back(ib[e]);
      g.meas.insert(g.meas.end(), meas + 3 * (size_t)e, meas + 3 * (size_t)e + 3);
      const double* inf = info6 ? info6 + 6 * (size_t)e : unit_info;
      g.info.insert(g.info.end(), inf, inf + 6);
      g.kind.push_back((uint8_t)k);
      g.n_kind[k]++;
    }
  *out = pg.release();
  return PGO_OK;
}

void pgo_graph_free(pgo_graph* g) { delete g; }

int32_t pgo_graph_num_poses(const pgo_graph* g) { return g ? g->g.n_poses() : 0; }
int32_t pgo_graph_num_edges(const pgo_graph* g) { return g ? g->g.n_edges() : 0; }
int32_t pgo_graph_num_edges_of_kind(const pgo_graph* g, int kind) {
  return (g && kind >= 0 && kind <= 2) ? g->g.n_kind[kind] : 0;
}
const int32_t* pgo_graph_pose_ids(const pgo_graph* g) { return g ? g->g.pose_id.data() : nullptr; }
double* pgo_graph_poses(pgo_graph* g) { return g ? g->g.pose.data() : nullptr; }
const int32_t* pgo_graph_edge_a(const pgo_graph* g) { return g ? g->g.ea.data() : nullptr; }
const int32_t* pgo_graph_edge_b(const pgo_graph* g) { return g ? g->g.eb.data() : nullptr; }
const double* pgo_graph_edge_meas(const pgo_graph* g) { return g ? g->g.meas.data() : nullptr; }
const double* pgo_graph_edge_info(const pgo_graph* g) { return g ? g->g.info.data() : nullptr; }
const uint8_t* pgo_graph_edge_kind(const pgo_graph* g) { return g ? g->g.kind.data() : nullptr; }

int pgo_inject_outliers(pgo_graph* pg, int32_t count, int64_t seed) {
  if (!pg || count < 0) return fail(PGO_ERR_INVALID_ARG, "pgo_inject_outliers: bad argument");
  Graph& g = pg->g;
  int32_t MAX = g.n_poses();
  if (MAX < 2 && count > 0) return fail(PGO_ERR_INVALID_ARG, "need at least 2 poses to add loops");
  // main.cpp:43 seeds once per run with time(0); a caller-chosen seed makes the
  // augmented graph reproducible
  srand(seed >= 0 ? (unsigned)seed : (unsigned)time(nullptr));
  const double info[6] = {2.0, 0.0, 0.0, 300.0, 0.0, 300.0};  // g2o_util.h:168
  for (int32_t i = 0; i < count; ++i) {
    int a = rand() % MAX;  // g2o_util.h:158-159: a first, then b
    int b = rand() % MAX;
    if (a == b) b = (b + 1) % MAX;
    // g2o_util.h:166: rand()/RAND_MAX is int/int => 0 unless rand() == RAND_MAX.
    // C++ leaves the evaluation order of the three arguments unspecified; the
    // three values are exchangeable in distribution and all 0 in practice.
    double m[3];
    m[0] = (double)(rand() / RAND_MAX);
    m[1] = (double)(rand() / RAND_MAX);
    m[2] = (double)(rand() / RAND_MAX);
    g.insert_edge(a, b, m, info, PGO_EDGE_BOGUS);
  }
  return PGO_OK;
}

static void print_double(FILE* f, double v, int precision) {
  if (precision <= 0) fprintf(f, "%g", v);  // default ostream formatting = %g, 6 significant digits
  else fprintf(f, "%.*g", precision, v);
}

int pgo_write_nodes(const pgo_graph* pg, const char* path, int precision) {
  if (!pg || !path) return fail(PGO_ERR_INVALID_ARG, "pgo_write_nodes: null argument");
  FILE* f = fopen(path, "w");
  if (!f) return fail(PGO_ERR_IO, std::string("cannot open ") + path + ": " + strerror(errno));
  const Graph& g = pg->g;
  for (int32_t i = 0; i < g.n_poses(); ++i) {  // g2o_util.h:98-101: "<index> <x> <y> <theta>"
    fprintf(f, "%d ", g.pose_id[i]);
    print_double(f, g.pose[3 * (size_t)i + 0], precision);
    fputc(' ', f);
    print_double(f, g.pose[3 * (size_t)i + 1], precision);
    fputc(' ', f);
    print_double(f, g.pose[3 * (size_t)i + 2], precision);
    fputc('\n', f);
  }
  bool bad = ferror(f);
  if (fclose(f) != 0 || bad) return fail(PGO_ERR_IO, std::string("write error on ") + path);
  return PGO_OK;
}

int pgo_write_edges(const pgo_graph* pg, const char* path) {
  if (!pg || !path) return fail(PGO_ERR_INVALID_ARG, "pgo_write_edges: null argument");
  FILE* f = fopen(path, "w");
  if (!f) return fail(PGO_ERR_IO, std::string("cannot open ") + path + ": " + strerror(errno));
  const Graph& g = pg->g;
  // g2o_util.h:109-111,184: odometry, closure, bogus; "<a.index> <b.index> <type>"
  for (int32_t e = 0; e < g.n_edges(); ++e)
    fprintf(f, "%d %d %d\n", g.pose_id[g.ea[e]], g.pose_id[g.eb[e]], (int)g.kind[e]);
  bool bad = ferror(f);
  if (fclose(f) != 0 || bad) return fail(PGO_ERR_IO, std::string("write error on ") + path);
  return PGO_OK;
}

int pgo_write_switches(const pgo_graph* pg, const char* path, const double* sw) {
  if (!pg || !path || !sw) return fail(PGO_ERR_INVALID_ARG, "pgo_write_switches: null argument");
  FILE* f = fopen(path, "w");
  if (!f) return fail(PGO_ERR_IO, std::string("cannot open ") + path + ": " + strerror(errno));
  const Graph& g = pg->g;
  // g2o_util.h:122-146: "<a> <b> <type> <prior> <value>", default ostream formatting (%g), one section per list
  static const char* head[3] = {"Odometry EDGES AHEAD\n", "Closure EDGES AHEAD\n", "BOGUS EDGES AHEAD\n"};
  int32_t e = 0;
  for (int k = 0; k < 3; ++k) {
    fputs(head[k], f);
    for (int32_t i = 0; i < g.n_kind[k]; ++i, ++e) {
      fprintf(f, "%d %d %d %g ", g.pose_id[g.ea[e]], g.pose_id[g.eb[e]], (int)g.kind[e], 1.0);
      fprintf(f, "%g\n", k == 0 ? 1.0 : sw[e]);
    }
  }
  bool bad = ferror(f);
  if (fclose(f) != 0 || bad) return fail(PGO_ERR_IO, std::string("write error on ") + path);
  return PGO_OK;
}

int pgo_write_g2o(const pgo_graph* pg, const char* path) {
  if (!pg || !path) return fail(PGO_ERR_INVALID_ARG, "pgo_write_g2o: null argument");
  FILE* f = fopen(path, "w");
  if (!f) return fail(PGO_ERR_IO, std::string("cannot open ") + path + ": " + strerror(errno));
  const Graph& g = pg->g;
  for (int32_t i = 0; i < g.n_poses(); ++i)
    fprintf(f, "VERTEX_SE2 %d %.17g %.17g %.17g\n", g.pose_id[i], g.pose[3 * (size_t)i], g.pose[3 * (size_t)i + 1],
            g.pose[3 * (size_t)i + 2]);
  for (int32_t e = 0; e < g.n_edges(); ++e) {
    const double* m = &g.meas[3 * (size_t)e];
    const double* q = &g.info[6 * (size_t)e];
    fprintf(f, "EDGE_SE2 %d %d %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", g.ea[e], g.eb[e], m[0], m[1],
            m[2], q[0], q[1], q[2], q[3], q[4], q[5]);
  }
  bool bad = ferror(f);
  if (fclose(f) != 0 || bad) return fail(PGO_ERR_IO, std::string("write error on ") + path);
  return PGO_OK;
}

int pgo_synth_manhattan(int32_t n_poses, double edges_per_pose, double outlier_frac, uint64_t seed, pgo_graph** out) {
  if (!out) return fail(PGO_ERR_INVALID_ARG, "pgo_synth_manhattan: null argument");
  std::unique_ptr<pgo_graph> pg(new pgo_graph);
  int st = pgo::synth_manhattan(n_poses, edges_per_pose, outlier_frac, seed, &pg->g);
  if (st != PGO_OK) return st;
  *out = pg.release();
  return PGO_OK;
}

}  // extern "C"
