#!/bin/bash
# builds the tree-step profiling variant of the library (same sources, -DAZR_TREE_PROF on azr_engine.hip only) into csrc/dbg/
# (git-ignored; run here, the .so travels to the GPU box), then on the box:
#   AZR_EXP_LIB=alphazero-risk_amd/csrc/dbg/libazr_prof.so python tools/tree_prof.py
set -e
cd "$(dirname "$0")/../alphazero-risk_amd/csrc"
make -s
mkdir -p dbg
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -w -DAZR_TREE_PROF -c azr_engine.hip -o dbg/azr_engine_prof.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o dbg/libazr_prof.so dbg/azr_engine_prof.o azr_net.o azr_net_bf16.o azr_tower_sb.o azr_tower_sc.o azr_tower_fx.o azr_train.o
ls -la dbg/libazr_prof.so
