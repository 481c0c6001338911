#include "solver.hpp"
namespace kin {
struct Solver {};
}
kin_network::kin_network() {}
kin_network::~kin_network() {
  solver.reset();
  if (stream) (void)hipStreamDestroy(stream);
}
namespace kin {
int solve_entry(kin_network*, const kin_params&, const double*, const double*, const double*, const double*, int64_t, kin_stats*) {
  throw KinError(ERR_STATE, "solver not built yet");
}
void solution_max(kin_network*, double*) { throw KinError(ERR_STATE, "solver not built yet"); }
}
