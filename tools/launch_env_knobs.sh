#!/bin/bash
# run ON the GPU box: tools/launch_env_knobs.py under one runtime knob at a time (twice: the order must not matter)
cd "$(dirname "$0")/.."
for rep in 1 2; do
  python tools/launch_env_knobs.py
  HIP_FORCE_DEV_KERNARG=0 python tools/launch_env_knobs.py
  HIP_FORCE_DEV_KERNARG=1 python tools/launch_env_knobs.py
  AMD_OPT_FLUSH=0 python tools/launch_env_knobs.py
  AMD_OPT_FLUSH=1 python tools/launch_env_knobs.py
  DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 python tools/launch_env_knobs.py
  DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 python tools/launch_env_knobs.py
  AMD_DIRECT_DISPATCH=0 python tools/launch_env_knobs.py
  DEBUG_HIP_GRAPH_BATCH_SIZE=64 python tools/launch_env_knobs.py
done
