// Kernel instantiations: D1Q3, double.
#define LT_S lt::D1Q3
#define LT_T double
#define LT_TAG d1q3_f64
#define LT_HAS_KBC 0
#define LT_IS_3D 0
#include "unit.inc"
