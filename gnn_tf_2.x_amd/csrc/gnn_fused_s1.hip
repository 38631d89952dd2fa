// Split-arithmetic (fp32 as three bf16 pieces on the bf16 MFMA, impl 2) instantiations of the fused iteration kernel for
// net_state with 1 Dense layer(s).
#include "gnn_fused_kernel.h"

bool gnn_fused_launch_s1(int act, int nt, int ntl, const GnnFusedArgs &a, unsigned grid, size_t lds_bytes, hipStream_t st)
{
    return gnn_fused_dev::launch_act<1, true>(act, nt, ntl, a, grid, lds_bytes, st);
}
