mkdir -p gpurun_out
python tools/zoo_diag.py 16000 1940003 3 5 16000 1940053 3 5 2>&1 | grep -v amdgpu.ids > gpurun_out/diag_imp.txt
cat gpurun_out/diag_imp.txt
