// Library-owned device scratch (block sums for the prefix-sum ray compaction, MLP wgrad partials, the records of the hash-grid
// backward).  One buffer per (device, STREAM, slot); grows monotonically.  The kernels of one C-ABI call that share a slot are
// enqueued on the call's stream, and the next call on that stream is ordered behind them -- so calls on different streams (or from
// different threads, each on its own stream) never share scratch.  Two host threads launching on the SAME stream are the caller's
// to serialise, as with any stream.
#pragma once
#include "common.hpp"

namespace nerftex {

enum WorkspaceSlot { kWsMarch = 0, kWsCompact = 1, kWsMlp = 2, kWsGrid = 3, kWsGridFwd = 4, kWsGridBins = 5, kWsOccupancy = 6, kWsOccupancyList = 7, kWsMlpB = 8, kWsSlots = 9 };

// returns nullptr (and sets the error text) on allocation failure
void* workspace(WorkspaceSlot slot, size_t bytes, hipStream_t stream);
void release_workspaces();

}  // namespace nerftex
