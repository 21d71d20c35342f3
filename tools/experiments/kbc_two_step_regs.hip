// Round 3, VERDICT r02 item 6: what does KBC cost inside the masked two-step kernel?  Compile only:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../../include -I../../lettuce_amd/csrc -Rpass-analysis=kernel-resource-usage -c kbc_two_step_regs.hip
#include "twostep_masked.hpp"
template __global__ void lt::lbm2m_kernel<float, lt::D3Q27, 0, 2, 64, 4, 2>(const lt::KParams<float>, const int);
template __global__ void lt::lbm2m_kernel<float, lt::D3Q27, 0, 1, 64, 4, 2>(const lt::KParams<float>, const int);
