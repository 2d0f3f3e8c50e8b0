// Words of a context's host-mapped error area (range_hip.hip: range_ctx::h_async_err), set to 1 - a
// plain system-scope store - by a persistent kernel whose bounded in-launch wait for other
// workgroups gave up.
#pragma once
#include <stdint.h>

namespace range_hip {
constexpr int RANGE_ASYNC_WORD_ENCODER = 0;   // encoder_tile_kernel: the tile's rows were written as NaN
constexpr int RANGE_ASYNC_WORD_TOPK = 1;      // topks_tail: the query's results were written as NaN / -1
}  // namespace range_hip
