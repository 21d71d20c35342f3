// Kernel instantiations: D3Q19, double.
#define LT_S lt::D3Q19
#define LT_T double
#define LT_TAG d3q19_f64
#define LT_HAS_KBC 0
#define LT_IS_3D 1
#include "unit.inc"
