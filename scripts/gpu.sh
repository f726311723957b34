#!/bin/bash
# Build libadaface_hip.so here (hipcc cross-compiles gfx950), then run a command on an MI355X box through gpurun.
#   scripts/gpu.sh [--timeout S] -- '<command>'
set -e
cd "$(dirname "$0")/.."
python -m adaface_amd.build >/dev/null
exec /usr/local/graft/bin/gpurun "$@"
