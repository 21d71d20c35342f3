// Lattice constants and the logical-axis -> memory-axis map.
//
// Velocity order, weights and opposite indices restate
//   lettuce/ext/_stencil/d1q3.py:8-10, d2q9.py:8-10, d3q15.py:8-13, d3q19.py:8-13, d3q27.py:8-12
// (reference, /root/reference).  The order matters: it is the q index of the
// user-visible tensor f[q, ...].
#pragma once
#include <utility>

namespace lt {

struct D1Q3 {
  static constexpr int D = 1, Q = 3;
  static constexpr const char *NAME = "d1q3";
  static constexpr int E[3][3] = {{0, 0, 0}, {1, 0, 0}, {-1, 0, 0}};
  static constexpr double W[3] = {2.0 / 3.0, 1.0 / 6.0, 1.0 / 6.0};
  static constexpr int OPP[3] = {0, 2, 1};
};

struct D2Q9 {
  static constexpr int D = 2, Q = 9;
  static constexpr const char *NAME = "d2q9";
  static constexpr int E[9][3] = {{0, 0, 0}, {1, 0, 0}, {0, 1, 0}, {-1, 0, 0}, {0, -1, 0},
                                  {1, 1, 0}, {-1, 1, 0}, {-1, -1, 0}, {1, -1, 0}};
  static constexpr double W[9] = {4.0 / 9.0, 1.0 / 9.0, 1.0 / 9.0, 1.0 / 9.0, 1.0 / 9.0,
                                  1.0 / 36.0, 1.0 / 36.0, 1.0 / 36.0, 1.0 / 36.0};
  static constexpr int OPP[9] = {0, 3, 4, 1, 2, 7, 8, 5, 6};
};

struct D3Q15 {
  static constexpr int D = 3, Q = 15;
  static constexpr const char *NAME = "d3q15";
  static constexpr int E[15][3] = {
      {0, 0, 0},  {1, 0, 0},  {-1, 0, 0},  {0, 1, 0},  {0, -1, 0}, {0, 0, 1},   {0, 0, -1}, {1, 1, 1},
      {-1, -1, -1}, {1, 1, -1}, {-1, -1, 1}, {1, -1, 1}, {-1, 1, -1}, {1, -1, -1}, {-1, 1, 1}};
  static constexpr double W[15] = {2.0 / 9.0,  1.0 / 9.0,  1.0 / 9.0,  1.0 / 9.0,  1.0 / 9.0,
                                   1.0 / 9.0,  1.0 / 9.0,  1.0 / 72.0, 1.0 / 72.0, 1.0 / 72.0,
                                   1.0 / 72.0, 1.0 / 72.0, 1.0 / 72.0, 1.0 / 72.0, 1.0 / 72.0};
  static constexpr int OPP[15] = {0, 2, 1, 4, 3, 6, 5, 8, 7, 10, 9, 12, 11, 14, 13};
};

struct D3Q19 {
  static constexpr int D = 3, Q = 19;
  static constexpr const char *NAME = "d3q19";
  static constexpr int E[19][3] = {
      {0, 0, 0},  {1, 0, 0},  {-1, 0, 0}, {0, 1, 0},  {0, -1, 0},  {0, 0, 1},  {0, 0, -1},
      {0, 1, 1},  {0, -1, -1}, {0, 1, -1}, {0, -1, 1}, {1, 0, 1},  {-1, 0, -1}, {1, 0, -1},
      {-1, 0, 1}, {1, 1, 0},  {-1, -1, 0}, {1, -1, 0}, {-1, 1, 0}};
  static constexpr double W[19] = {
      1.0 / 3.0,  1.0 / 18.0, 1.0 / 18.0, 1.0 / 18.0, 1.0 / 18.0, 1.0 / 18.0, 1.0 / 18.0,
      1.0 / 36.0, 1.0 / 36.0, 1.0 / 36.0, 1.0 / 36.0, 1.0 / 36.0, 1.0 / 36.0, 1.0 / 36.0,
      1.0 / 36.0, 1.0 / 36.0, 1.0 / 36.0, 1.0 / 36.0, 1.0 / 36.0};
  static constexpr int OPP[19] = {0, 2, 1, 4, 3, 6, 5, 8, 7, 10, 9, 12, 11, 14, 13, 16, 15, 18, 17};
};

struct D3Q27 {
  static constexpr int D = 3, Q = 27;
  static constexpr const char *NAME = "d3q27";
  static constexpr int E[27][3] = {
      {0, 0, 0},   {1, 0, 0},  {-1, 0, 0},  {0, 1, 0},   {0, -1, 0}, {0, 0, 1},   {0, 0, -1},
      {0, 1, 1},   {0, -1, -1}, {0, 1, -1}, {0, -1, 1},  {1, 0, 1},  {-1, 0, -1}, {1, 0, -1},
      {-1, 0, 1},  {1, 1, 0},  {-1, -1, 0}, {1, -1, 0},  {-1, 1, 0}, {1, 1, 1},   {-1, -1, -1},
      {1, 1, -1},  {-1, -1, 1}, {1, -1, 1}, {-1, 1, -1}, {1, -1, -1}, {-1, 1, 1}};
  static constexpr double W[27] = {
      8.0 / 27.0,  2.0 / 27.0,  2.0 / 27.0,  2.0 / 27.0,  2.0 / 27.0,  2.0 / 27.0,  2.0 / 27.0,
      1.0 / 54.0,  1.0 / 54.0,  1.0 / 54.0,  1.0 / 54.0,  1.0 / 54.0,  1.0 / 54.0,  1.0 / 54.0,
      1.0 / 54.0,  1.0 / 54.0,  1.0 / 54.0,  1.0 / 54.0,  1.0 / 54.0,  1.0 / 216.0, 1.0 / 216.0,
      1.0 / 216.0, 1.0 / 216.0, 1.0 / 216.0, 1.0 / 216.0, 1.0 / 216.0, 1.0 / 216.0};
  static constexpr int OPP[27] = {0,  2,  1,  4,  3,  6,  5,  8,  7,  10, 9,  12, 11, 14,
                                  13, 16, 15, 18, 17, 20, 19, 22, 21, 24, 23, 26, 25};
};

// Memory axes a0 (fastest), a1, a2 of a population field [q][a2][a1][a0].
//   LAYOUT 0 (reference layout): 3-D a0=z a1=y a2=x; 2-D a0=y a1=x (a2 has extent 1);
//                                1-D a0=x (a1, a2 have extent 1).
//   LAYOUT 1 (slab layout, 3-D): a0=x a1=y a2=z.
template <class S, int LAYOUT>
struct MemMap {
  static constexpr int logical(int m) {
    return S::D == 1 ? m
                     : (S::D == 2 ? (m == 0 ? 1 : (m == 1 ? 0 : 2)) : (LAYOUT == 0 ? 2 - m : m));
  }
  static constexpr int memory(int logical_axis) {
    return S::D == 1 ? logical_axis
                     : (S::D == 2 ? (logical_axis == 0 ? 1 : (logical_axis == 1 ? 0 : 2))
                                  : (LAYOUT == 0 ? 2 - logical_axis : logical_axis));
  }
  static constexpr int e(int q, int m) { return S::E[q][logical(m)]; }
};

// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(<N-1>)
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
  static_for_impl(static_cast<F &&>(f), std::make_integer_sequence<int, N>{});
}

}  // namespace lt
