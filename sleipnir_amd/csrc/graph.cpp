#include "graph.hpp"

namespace slpx {

Graph& graph() {
  static thread_local Graph g;
  return g;
}

}  // namespace slpx
