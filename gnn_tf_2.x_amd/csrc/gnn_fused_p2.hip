// Wave-pair form of the fused iteration kernel (gnn_fused_pair_kernel.h: split arithmetic, state width 64, 128-wide hidden layers),
// net_state with 2 Dense layers.
#include "gnn_fused_pair_kernel.h"

bool gnn_fused_launch_p2(int act, const GnnFusedArgs &a, unsigned grid, size_t lds_bytes, hipStream_t st)
{
    return gnn_fused_dev::launch_pair_act<2>(act, a, grid, lds_bytes, st);
}
