mkdir -p gpurun_out
timeout 1500 python tools/long_utterance_probe.py 16000 300 2>&1 | grep -v amdgpu.ids > gpurun_out/long.txt
cat gpurun_out/long.txt
