// Error text and version of libprl_hip.so.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include "prl_common.h"

namespace prl {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

void clear_error() { g_err[0] = '\0'; }

}  // namespace prl

extern "C" const char* prl_last_error(void) { return prl::g_err; }

extern "C" int prl_abi_version(void) { return 1; }

// Walks a captured graph (child graphs included) and counts its nodes and, among them, the memset nodes.
static int count_nodes(hipGraph_t graph, int64_t* total, int64_t* memsets, int depth) {
  if (depth > 8) return PRL_OK;  // torch captures are flat; a bound keeps a malformed graph from recursing for ever
  size_t n = 0;
  if (hipGraphGetNodes(graph, nullptr, &n) != hipSuccess) return PRL_ELAUNCH;
  if (n == 0) return PRL_OK;
  hipGraphNode_t* nodes = (hipGraphNode_t*)malloc(n * sizeof(hipGraphNode_t));
  if (!nodes) return PRL_ELAUNCH;
  int rc = PRL_OK;
  if (hipGraphGetNodes(graph, nodes, &n) != hipSuccess) rc = PRL_ELAUNCH;
  for (size_t i = 0; rc == PRL_OK && i < n; ++i) {
    hipGraphNodeType type;
    if (hipGraphNodeGetType(nodes[i], &type) != hipSuccess) {
      rc = PRL_ELAUNCH;
      break;
    }
    *total += 1;
    if (type == hipGraphNodeTypeMemset) *memsets += 1;
    if (type == hipGraphNodeTypeGraph) {
      hipGraph_t child;
      if (hipGraphChildGraphNodeGetGraph(nodes[i], &child) != hipSuccess) rc = PRL_ELAUNCH;
      else rc = count_nodes(child, total, memsets, depth + 1);
    }
  }
  free(nodes);
  return rc;
}

extern "C" int prl_graph_node_counts(void* graph, int64_t* n_nodes, int64_t* n_memset_nodes) {
  PRL_REQUIRE(graph && n_nodes && n_memset_nodes, "prl_graph_node_counts: null argument");
  *n_nodes = *n_memset_nodes = 0;
  int rc = count_nodes((hipGraph_t)graph, n_nodes, n_memset_nodes, 0);
  if (rc != PRL_OK) {
    (void)hipGetLastError();
    prl::set_error("prl_graph_node_counts: the HIP runtime refused a graph query");
  }
  return rc;
}
