// Kernel instantiations: D3Q15, float.
#define LT_S lt::D3Q15
#define LT_T float
#define LT_TAG d3q15_f32
#define LT_HAS_KBC 0
#define LT_IS_3D 1
#include "unit.inc"
