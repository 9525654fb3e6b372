mkdir -p gpurun_out
{
ZOO2=1 python tools/raw_diag.py 24000 1800012 3 2>&1 | grep -v amdgpu.ids

} > gpurun_out/diag_stairs2.txt 2>&1
grep -A40 "base frames" gpurun_out/diag_stairs2.txt
