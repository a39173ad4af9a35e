// Library-owned device scratch (block sums for the prefix-sum ray compaction, MLP wgrad partials, the records of the hash-grid
// backward).  One buffer per (device, STREAM, slot); grows monotonically.  The kernels of one C-ABI call that share a slot are
// enqueued on the call's stream, and the next call on that stream is ordered behind them -- so calls on different streams (or from
// different threads, each on its own stream) never share scratch.  Two host threads launching on the SAME stream are the caller's
// to serialise, as with any stream.
//
// STREAM CAPTURES share ONE set per device (a capturing stream is a one-off handle; what it records is replayed later, anywhere): every
// captured launch bakes in a pointer into that set.  The invariant a caller of replayed graphs has to keep -- and the only thing that makes
// two graphs safe to replay side by side:
//   * two graphs may run CONCURRENTLY only if the slots their kernels use are DISJOINT.  By purpose: the training march (march_rays_train*)
//     uses kWsMarch and nothing else; one training step behind it (field forward / backward, hash-grid backward, compositing) uses
//     kWsMlp, kWsMlpB, kWsGridFwd, kWsGrid, kWsGridBins; the inference loop kWsCompact; the occupancy update kWsOccupancy*.
//     So "the march of the next step beside this step" (bench.py, ngp_harness/accelerate.py) is safe, two steps side by side are not, two
//     marches side by side are not (they are replayed in order on one side stream);
//   * within one slot the replays must be ordered (same stream, or event-ordered): the nerftex_grid_encode_backward_phase calls of one
//     step read the scratch their phase-1 call left, in that order.
// tests/test_gpu_round4.py::test_concurrent_graphs_use_disjoint_scratch_slots holds the two graph kinds to this list.
#pragma once
#include <atomic>
#include "common.hpp"

namespace nerftex {

enum WorkspaceSlot { kWsMarch = 0, kWsCompact = 1, kWsMlp = 2, kWsGrid = 3, kWsGridFwd = 4, kWsGridBins = 5, kWsOccupancy = 6, kWsOccupancyList = 7, kWsMlpB = 8, kWsSlots = 9 };

// returns nullptr (and sets the error text) on allocation failure
void* workspace(WorkspaceSlot slot, size_t bytes, hipStream_t stream);
void release_workspaces();
// which slots have been handed out for `stream` (a capturing stream: the shared capture set) since the last reset: bit s = slot s (tests)
extern std::atomic<unsigned> g_ws_touched;  // (callers on different threads use different per-stream scratch: the mask is the one shared word)

}  // namespace nerftex
