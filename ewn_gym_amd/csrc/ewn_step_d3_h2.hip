// ewn_step_d3_h2.hip -- k_step_d3 on the 'two_min_dist' table image (envs/minimax_ewn.py:133-178): the same fused step and search,
// a side's leaf index read from its mask's two highest bits (template parameter H2 of d3_search / d5_search).
#define D3_H2 1
#define D3_LAUNCHER ewn_launch_step_d3_h2
#include "ewn_step_d3_tu.inc"
