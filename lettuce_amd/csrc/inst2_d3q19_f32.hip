// Kernel instantiations: D3Q19, float.  Part 2: the unmasked two-step launches (unit.inc, LT_PART).
#define LT_S lt::D3Q19
#define LT_T float
#define LT_TAG d3q19_f32
#define LT_HAS_KBC 0
#define LT_IS_3D 1
#define LT_PART 2
#include "unit.inc"
