// EXPERIMENT (diagnostic build only, round 3): 16-node-tile instantiations (gnn_fused16_kernel.h) of the fused iteration kernel for net_state with 2 Dense layers.
#include "experiments/gnn_fused16_kernel.h"

static_assert(gnn_fused_dev::GNN_F16_WAVES == GNN_FUSED16_WAVES, "host and device disagree on the workgroup size");

bool gnn_fused_launch_h2(int act, int nf, const GnnFusedArgs &a, unsigned grid, size_t lds_bytes, hipStream_t st)
{
    return gnn_fused_dev::launch16_act<2>(act, nf, a, grid, lds_bytes, st);
}
