// Kernel instantiations: D3Q27, double.
#define LT_S lt::D3Q27
#define LT_T double
#define LT_TAG d3q27_f64
#define LT_HAS_KBC 1
#define LT_IS_3D 1
#include "unit.inc"
