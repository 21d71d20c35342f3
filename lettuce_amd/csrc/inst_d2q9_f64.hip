// Kernel instantiations: D2Q9, double.
#define LT_S lt::D2Q9
#define LT_T double
#define LT_TAG d2q9_f64
#define LT_HAS_KBC 1
#define LT_IS_3D 0
#include "unit.inc"
