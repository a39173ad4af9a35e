// Library-owned device scratch (block sums for the prefix-sum ray compaction, MLP wgrad partials).
// One buffer per (device, slot); grows monotonically.  Work that uses a slot is stream-ordered by
// the caller: the kernels of one C-ABI call that share the slot are enqueued on the same stream, and
// two calls racing on DIFFERENT streams for the SAME slot must be serialised by the caller (the
// Python host side always uses torch's current stream).
#pragma once
#include "common.hpp"

namespace nerftex {

enum WorkspaceSlot { kWsMarch = 0, kWsCompact = 1, kWsMlp = 2, kWsGrid = 3, kWsGridFwd = 4, kWsGridBins = 5, kWsOccupancy = 6, kWsOccupancyList = 7, kWsSlots = 8 };

// returns nullptr (and sets the error text) on allocation failure
void* workspace(WorkspaceSlot slot, size_t bytes);
void release_workspaces();

}  // namespace nerftex
