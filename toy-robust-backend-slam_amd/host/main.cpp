// Command-line front end with the reference's run contract (DCS-ceres/main.cpp:22-40, do_build.sh:10):
//     ./main DATASET NUM_OUTLIER_LOOPS METHOD [--seed S] [--data DIR] [--save DIR] [--device D] [--precision P]
// METHOD 0 = baseline, 1 = DCS, 2 = switchable constraints run on the MI355X backend; 3/4 are outside it (exit code 3).
// Reads DIR/DATASET.g2o (default ../data, as the reference), writes init_/opt_ nodes+edges text files that
// drawer/plot_results.py consumes (default ../save; unlike the reference the directory is created).
#include <sys/stat.h>

#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#include "ceres_error.h"
#include "g2o_util.h"
#include "graph.h"
#include "pgo_problem.h"

int main(int argc, char* argv[]) {
  if (argc < 4) {
    std::cout << "Usage: " << argv[0] << " DATASET NUM_OUTLIER_LOOPS METHOD [--seed S] [--data DIR] [--save DIR] [--device D]\n"
              << "METHOD: 0=baseline, 1=DCS, 2=Switchable (3=Layer, 4=Simple Layer MCTS: not in this backend)\n"
              << "Example: " << argv[0] << " INTEL 50 1\n";
    return -1;
  }
  std::string base = "../data", save = "../save";
  long long seed = -1;
  int device = 0, precision = 0;
  for (int i = 4; i + 1 < argc; i += 2) {
    if (!strcmp(argv[i], "--seed")) seed = atoll(argv[i + 1]);
    else if (!strcmp(argv[i], "--data")) base = argv[i + 1];
    else if (!strcmp(argv[i], "--save")) save = argv[i + 1];
    else if (!strcmp(argv[i], "--device")) device = atoi(argv[i + 1]);
    else if (!strcmp(argv[i], "--precision")) precision = atoi(argv[i + 1]);
    else { std::cerr << "unknown option " << argv[i] << "\n"; return -1; }
  }
  const int method = atoi(argv[3]);
  if (method < 0 || method > 2) {
    std::cerr << "METHOD " << method << " is not part of the MI355X backend (METHOD 0, 1 and 2 only)\n";
    return 3;
  }
  try {
    std::cout << "Start Reading PoseGraph\n";
    ReadG2O g2o_manager(base + "/" + argv[1] + ".g2o");
    g2o_manager.add_random_C(atoi(argv[2]), seed);
    mkdir(save.c_str(), 0755);
    g2o_manager.writePoseGraph_nodes(save + "/init_nodes.txt", precision);
    g2o_manager.writePoseGraph_edges(save + "/init_edges.txt");
    std::cout << "total nodes : " << g2o_manager.nNodes.size() << std::endl;
    std::cout << "total nEdgesOdometry : " << g2o_manager.nEdgesOdometry.size() << std::endl;
    std::cout << "total nEdgesClosure : " << g2o_manager.nEdgesClosure.size() << std::endl;
    std::cout << "total nEdgesBogus : " << g2o_manager.nEdgesBogus.size() << std::endl;

    pgo::Problem problem;
    pgo::LossFunction* loss_function = new pgo::HuberLoss(0.01);
    const bool DCS_ON = (method == 1), SC_ON = (method == 2);
    std::vector<double> switch_priors;      // for SC (reference main.cpp:105-107)
    std::vector<double*> switch_variables;
    const double sc_prior_lambda = 1.0;
    for (Edge* ed : g2o_manager.nEdgesOdometry)
      problem.AddResidualBlock(OdometryResidue::Create(ed->x, ed->y, ed->theta), loss_function, ed->a->p, ed->b->p);
    for (auto* list : {&g2o_manager.nEdgesClosure, &g2o_manager.nEdgesBogus})
      for (Edge* ed : *list) {
        if (SC_ON) {
          double* s = new double(1.0);
          switch_variables.push_back(s);
          switch_priors.push_back(1.0);
          problem.AddResidualBlock(SwitchableClosureResidue::Create(ed->x, ed->y, ed->theta), loss_function, ed->a->p, ed->b->p, s);
          problem.AddResidualBlock(SwitchPriorResidue::Create(sc_prior_lambda), nullptr, s);
        } else {
          problem.AddResidualBlock(DCS_ON ? DCSClosureResidue::Create(ed->x, ed->y, ed->theta)
                                          : OdometryResidue::Create(ed->x, ed->y, ed->theta),
                                   loss_function, ed->a->p, ed->b->p);
        }
      }
    problem.SetParameterBlockConstant(g2o_manager.nNodes[0]->p);

    pgo::Solver::Options options;
    options.minimizer_progress_to_stdout = true;
    options.linear_solver_type = pgo::SPARSE_NORMAL_CHOLESKY;
    options.device = device;
    pgo::Solver::Summary summary;
    pgo::Solve(options, &problem, &summary);
    std::cout << summary.FullReport() << std::endl;
    delete loss_function;

    g2o_manager.writePoseGraph_nodes(save + "/opt_nodes.txt", precision);
    g2o_manager.writePoseGraph_edges(save + "/opt_edges.txt");
    if (SC_ON) g2o_manager.writePoseGraph_switches(save + "/switches.txt", switch_priors, switch_variables);
    for (double* s : switch_variables) delete s;
  } catch (const std::exception& e) {
    std::cerr << "error: " << e.what() << std::endl;
    return 2;
  }
  return 0;
}
