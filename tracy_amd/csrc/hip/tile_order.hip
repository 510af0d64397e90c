// tile_order.hip -- see tile_order.h
#include "tile_order.h"

#include <hipcub/hipcub.hpp>

#include <cstdlib>

namespace tile_order {
namespace {

constexpr uint32_t kFracBits = 8;   // most the key can hold; the default is chosen below

// cost class = a tiny float: the position of the leading one and the `frac_bits` bits after it
__global__ void k_cost_class(const uint32_t* __restrict__ cost, uint32_t n, uint32_t frac_bits, uint32_t* __restrict__ keys, uint32_t* __restrict__ iota) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t c = cost[i];
    uint32_t key = 0;
    if (c) {
        const uint32_t e = 31u - (uint32_t)__clz((int)c);                      // floor(log2(cost))
        const uint32_t frac = e >= frac_bits ? (c >> (e - frac_bits)) & ((1u << frac_bits) - 1u) : 0u;
        key = ((e + 1u) << frac_bits) | frac;
    }
    keys[i] = key;
    iota[i] = i;
}

}  // namespace

size_t scratch_bytes(uint32_t n) {
    size_t bytes = 0;
    (void)hipcub::DeviceRadixSort::SortPairsDescending(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr,
                                                       (int)n, 0, 6 + (int)kFracBits);
    return bytes;
}

int sort_by_cost(const uint32_t* cost, uint32_t* order, uint32_t n, uint32_t* keys, uint32_t* keys_sorted, uint32_t* iota, void* scratch, size_t scratch_size,
                 hipStream_t stream) {
    static const uint32_t frac_bits = [] { const char* e = getenv("TRH_TILE_CLASS_BITS"); return e ? (uint32_t)atoi(e) : kFracBits; }();   // development knob
    hipLaunchKernelGGL(k_cost_class, dim3((n + 255) / 256), dim3(256), 0, stream, cost, n, frac_bits <= kFracBits ? frac_bits : kFracBits, keys, iota);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    return (int)hipcub::DeviceRadixSort::SortPairsDescending(scratch, scratch_size, keys, keys_sorted, iota, order, (int)n, 0, 6 + (int)kFracBits, stream);
}

}  // namespace tile_order
