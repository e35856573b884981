// Tile geometry shared by the k=4, stride=2, pad=1 convolution kernels.
#pragma once
#include "pgan_common.h"
#include <utility>

namespace pgan {

// A tile of BN consecutive S-side pixels in (n, oh, ow) order: either R rows of one sample or
// SEG whole samples.
template <int HS, int BN>
struct PixTile {
    static constexpr int WS = HS, PIX = HS * HS;
    static constexpr int SEG = (BN >= PIX) ? BN / PIX : 1;
    static constexpr int R = (BN >= PIX) ? HS : BN / WS;
    static constexpr int TPS = (BN >= PIX) ? 1 : PIX / BN;
    static_assert(BN % WS == 0, "tile must hold whole rows");
    static_assert((BN >= PIX) ? (BN % PIX == 0) : (PIX % BN == 0), "tile/sample mismatch");
    __host__ __device__ static int ntiles(int N) { return (BN >= PIX) ? (N + SEG - 1) / SEG : N * TPS; }
    __device__ static void origin(int tile, int& n0, int& oh0) {
        if (BN >= PIX) { n0 = tile * SEG; oh0 = 0; }
        else { n0 = tile / TPS; oh0 = (tile % TPS) * R; }
    }
};

__device__ __forceinline__ int xcd_remap(int b, int total) {
    // blocks b and b+8 share an XCD (round-robin dispatch): give each XCD a contiguous run of
    // logical tiles so tiles that share an operand panel hit the same L2.  Speed only.
    return (total % 8 == 0) ? (b % 8) * (total / 8) + b / 8 : b;
}

template <int V> struct IC { static constexpr int value = V; };

// call f(IC<0>{}), f(IC<1>{}), ... f(IC<N-1>{}) in order
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    [&]<int... Is>(std::integer_sequence<int, Is...>) { (f(IC<Is>{}), ...); }(std::make_integer_sequence<int, N>{});
}

}  // namespace pgan
