// split_guard.hip -- host-side range check for consumers of split-K partial sums.
//
// Producers of K-split GEMM partials (gemm_lin.hip, skinny_dma.hip, wgrad_dma.hip, ...) hand a slab
// pointer and a split count to the kernel that sums them (lin_reduce_epilogue, the LSTM cell kernels,
// the attention kernels, the criterion head, splitk_reduce_acc).  Those consumers read
// slab[split][rows][cols] for split < nsplit with no bound of their own, so a stale count or offset
// would be an out-of-bounds device read (a GPU fault, not an error code).  Every consumer launcher
// therefore asks split_span_ok() first: the span must lie inside ONE workspace registered by its
// owner (rau_create registers the ctx's three slabs; tools register theirs) and the count must be in
// [1, kMaxSplits] -- otherwise the launcher returns hipErrorIllegalState and launches nothing, which
// the step-level callers report as RAU_ERR_STATE.
#include <algorithm>
#include <mutex>
#include <vector>

#include "kernels.h"

namespace rau {

namespace {
struct Ws { const float* base; size_t floats; };
std::mutex g_mu;
std::vector<Ws> g_ws;
}  // namespace

void split_ws_register(const float* base, size_t floats) {
  if (!base || !floats) return;
  std::lock_guard<std::mutex> lk(g_mu);
  for (Ws& w : g_ws)
    if (w.base == base) { w.floats = floats; return; }
  g_ws.push_back(Ws{base, floats});
}

void split_ws_unregister(const float* base) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_ws.erase(std::remove_if(g_ws.begin(), g_ws.end(), [&](const Ws& w) { return w.base == base; }),
             g_ws.end());
}

bool split_span_ok(const float* slab, long nsplit, size_t per_split) {
  if (nsplit == 0) return true;   // the consumer reads no partial at all
  if (!slab || nsplit < 0 || nsplit > kMaxSplits || per_split == 0) return false;
  std::lock_guard<std::mutex> lk(g_mu);
  for (const Ws& w : g_ws) {
    if (slab < w.base || slab >= w.base + w.floats) continue;
    const size_t off = (size_t)(slab - w.base);
    return (size_t)nsplit <= (w.floats - off) / per_split;
  }
  return false;
}

}  // namespace rau
