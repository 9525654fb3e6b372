#!/bin/bash
cp world_class_amd/libworldclass_hip.so /tmp/default.so
for v in /tmp/default.so tools/_ab/*.so; do
  cp $v world_class_amd/libworldclass_hip.so
  echo "== $v"
  python tools/microbench.py --stages ${1:-hcds} 2>&1 | grep -v "^fs=\|amdgpu.ids" 
done
cp /tmp/default.so world_class_amd/libworldclass_hip.so
