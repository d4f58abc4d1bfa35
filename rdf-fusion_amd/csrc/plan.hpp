// plan.hpp — compiled operator tree: DataSourceExec / FilterExec / HashJoinExec / CrossJoinExec /
// NestedLoopJoinExec / ProjectionExec over HBM-resident binding tables.
#pragma once
#include <vector>

#include "host_logic.hpp"
#include "kernels.hpp"
#include "store.hpp"

namespace rdfgpu {

struct SourceInfo {               // one DataSourceExec
  u32 node = 0;
  u32 components = RDFGPU_GSPO;   // chosen permutation (IndexPermutations::choose_index)
  ScanInstructions ix;            // instructions in the order of `components`
  PrunePlan prune;
  bool has_residual = false;      // predicates left after pruning => K2 compaction, else zero-copy slice
  u32 n_out = 0;
  u32 out_level[4] = {};          // index level feeding output column k (G,S,P,O order of first binding)
  u64 lo = 0, hi = 0;             // located range of the last execution
};

struct NodeInfo {
  rdfgpu_plan_node d;
  u32 width = 0;                  // output columns
  u32 n_proj = 0; u32 proj[kMaxCols] = {};
  ExprProgram prog{};             // filter / join filter
  int shape = 0;                  // filter kernel specialisation
  int source = -1;                // index into Plan::sources
};

struct BoundTable { std::vector<const u32*> cols; u64 n_rows = 0; bool bound = false; };

struct Plan {
  Store* store = nullptr;
  std::vector<NodeInfo> nodes;
  std::vector<SourceInfo> sources;
  std::vector<u32> pool;          // IN-set ids (host copy)
  u32* pool_dev = nullptr;        // same, on device
  u32 root = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev_start = nullptr, ev_stop = nullptr;
  std::vector<BoundTable> tables;

  // per-execution state
  std::vector<void*> allocs;      // pool blocks owned by the current result / intermediates
  u64* counters = nullptr;        // device u64 slots for operator output counts
  u32 counters_used = 0;
  LocateJob* jobs_dev = nullptr; u64* lohi_dev = nullptr;
  DevTable result; u64 result_rows = 0; bool executed = false;
  rdfgpu_metrics metrics{};
  // Arrow batch stream over a host copy of the result
  std::vector<std::vector<u32>> host_cols; bool host_valid = false; u64 cursor = 0;

  ~Plan();
  void execute();
  void ensure_host_copy();

 private:
  DevTable exec_node(u32 idx);
  DevTable exec_source(NodeInfo& nd);
  DevTable exec_filter(NodeInfo& nd);
  DevTable exec_join(NodeInfo& nd);
  void release_intermediates();
  template <class T> T* scratch(u64 n);
  u64* new_counter();
  u64 read_u64(const u64* dev);
};

Plan* plan_compile(Store* store, const rdfgpu_plan_desc* desc);

}  // namespace rdfgpu
