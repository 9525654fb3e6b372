mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 -o /tmp/cumask_probe tools/cumask_probe.hip && timeout 60 /tmp/cumask_probe > gpurun_out/cumask_probe.txt 2>&1
cat gpurun_out/cumask_probe.txt
