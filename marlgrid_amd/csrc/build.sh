#!/bin/bash
# Builds the engine for gfx950 (MI355X) in-tree.  hipcc cross-compiles without a GPU.
#   build.sh        -> libmarlgrid_hip.so      the product: no getenv, no measurement variants
#   build.sh ab     -> libmarlgrid_hip_ab.so   -DMG_AB_VARIANTS: env-switched A/B variants, loaded by tools/ only
# One object per translation unit, compiled in parallel (make -j); only changed files are rebuilt.
set -e
cd "$(dirname "$0")"
exec make -s -j8 "${1:-prod}"
