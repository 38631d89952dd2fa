// EXPERIMENT (diagnostic build only; measured slower than k_fused, profiles/r04_fused64_stamps.txt): 64-node-tile form of the fused iteration kernel (gnn_fused64_kernel.h), 2-layer net_state
#include "gnn_fused64_kernel.h"
bool gnn_fused_launch_w2(int act, const GnnFusedArgs &a, unsigned grid, size_t lds_bytes, hipStream_t st)
{
    return gnn_fused_dev::launch64_act<2>(act, a, grid, lds_bytes, st);
}
