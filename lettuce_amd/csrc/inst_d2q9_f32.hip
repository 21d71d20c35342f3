// Kernel instantiations: D2Q9, float.
#define LT_S lt::D2Q9
#define LT_T float
#define LT_TAG d2q9_f32
#define LT_HAS_KBC 1
#define LT_IS_3D 0
#include "unit.inc"
