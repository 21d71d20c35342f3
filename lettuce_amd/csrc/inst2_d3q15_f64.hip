// Kernel instantiations: D3Q15, double.  Part 2: the unmasked two-step launches (unit.inc, LT_PART).
#define LT_S lt::D3Q15
#define LT_T double
#define LT_TAG d3q15_f64
#define LT_HAS_KBC 0
#define LT_IS_3D 1
#define LT_PART 2
#include "unit.inc"
