#!/bin/bash
# A/B build of the fused kernel beside the product library: tools/ab_build.sh NAME "EXTRA FLAGS" [units...]
#   -> gnn_tf_2.x_amd/GNN/libgnn_hip_ab_NAME.so  (objects: csrc/obj_ab_NAME/; loaded through GNN_HIP_LIBRARY=<path>, never shipped).
# Only the listed translation units (default: the split-arithmetic instantiations gnn_fused_s1 s2 s3) are recompiled with the extra flags;
# every other object is the product's.
set -e
NAME=$1; EXTRA=$2; shift 2 || true
UNITS=${*:-gnn_fused_s1 gnn_fused_s2 gnn_fused_s3}
cd "$(dirname "$0")/../gnn_tf_2.x_amd/csrc"
make -j8 >/dev/null
mkdir -p obj_ab_$NAME
cp -p obj/*.o obj_ab_$NAME/
for u in $UNITS; do rm -f obj_ab_$NAME/$u.o; done
make -j8 OBJDIR=obj_ab_$NAME OUT=$PWD/../GNN/libgnn_hip_ab_$NAME.so EXTRA="$EXTRA" 2>&1 | grep -E "error|warning: |hazard" || true
ls -la ../GNN/libgnn_hip_ab_$NAME.so
