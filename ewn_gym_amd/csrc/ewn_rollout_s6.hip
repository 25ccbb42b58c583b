// one board size of the rollout kernel per translation unit (parallel build)
#define ROLL_S 6
#include "ewn_rollout_tu.inc"
