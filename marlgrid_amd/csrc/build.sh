#!/bin/bash
# Builds libmarlgrid_hip.so for gfx950 (MI355X) in-tree.  hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared \
    -I ../../include -I . \
    -Wall -Wno-unused-function \
    mg_api.hip mg_rng.hip mg_reset.hip mg_step.hip mg_render.hip mg_encode.hip mg_frame.hip \
    -o libmarlgrid_hip.so "$@"
echo "built $(pwd)/libmarlgrid_hip.so"
