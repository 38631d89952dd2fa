// Fused gather + MLP iteration kernel for gfx950 (placeholder until the kernel lands: the engine falls back to the
// per-op kernels of gnn_engine.hip).
#include "gnn_common.h"

bool gnn_fused_supported(const gnn_loop *) { return false; }
int gnn_fused_prepare(gnn_loop *) { return gnn_fail(GNN_ERR_UNSUPPORTED, "fused path not built"); }
int gnn_fused_pack(gnn_mlp *) { return GNN_OK; }
int gnn_fused_iteration(gnn_loop *, int) { return gnn_fail(GNN_ERR_UNSUPPORTED, "fused path not built"); }
void gnn_fused_release(gnn_mlp *) {}
