// One instantiation of the masked two-step kernel, compiled device-only to ISA by
// tests/test_host_api.py::test_masked_two_step_isa_has_no_dropped_register_copies (the kernel hipcc 7.2 miscompiled
// in round 3: csrc/Makefile header).
#include "twostep_masked.hpp"
template __global__ void lt::lbm2m_kernel<float, lt::D3Q27, 1, 1, 64, 4, 0>(const lt::KParams<float>, const int);
