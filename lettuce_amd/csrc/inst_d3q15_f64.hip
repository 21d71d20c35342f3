// Kernel instantiations: D3Q15, double.
#define LT_S lt::D3Q15
#define LT_T double
#define LT_TAG d3q15_f64
#define LT_HAS_KBC 0
#define LT_IS_3D 1
#include "unit.inc"
