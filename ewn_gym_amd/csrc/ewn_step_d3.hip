// ewn_step_d3.hip -- translation unit of the lean table-driven step kernel's instantiations (k_step_d3, ewn_step_d3.hpp):
// one launch = one env step.  Split from ewn_kernels.hip so the units compile in parallel.
#define D3_H2 0
#define D3_LAUNCHER ewn_launch_step_d3
#include "ewn_step_d3_tu.inc"
