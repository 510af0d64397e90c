// tile_order.h -- longest-processing-time-first ordering of the megakernel's tiles.
//
// The persistent waves of k_megakernel pull tiles from one counter. Tiles differ in cost by an order of
// magnitude (sky against the inside of the dragon), and a wave that draws an expensive tile near the end
// keeps the launch alive while the rest of the chip has drained: measured on the dragon, the mean wave
// lives 57 % as long as the longest one. A progressive renderer launches the same frame again and again,
// so every launch records what each tile cost (shader-clock ticks) and the next launch hands the tiles
// out in descending order of the binary logarithm of that cost -- raster order inside a cost class, so
// neighbouring tiles still run close in time and share cache lines. Results do not depend on the order:
// a tile is owned by one wave, and what happens inside a tile is the wave's own deterministic schedule.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tile_order {

// bytes of scratch sort_by_cost needs for n tiles (query once, allocate, pass to every call)
size_t scratch_bytes(uint32_t n);

// order[0..n) = tile indices sorted by descending cost class of cost[0..n); stable. Asynchronous on `stream`.
// keys / iota: n uint32 each (scratch the caller owns). Returns a hipError_t as int.
int sort_by_cost(const uint32_t* cost, uint32_t* order, uint32_t n, uint32_t* keys, uint32_t* keys_sorted, uint32_t* iota, void* scratch,
                 size_t scratch_size, hipStream_t stream);

}  // namespace tile_order
