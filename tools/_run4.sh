R=$PWD
cd /tmp; export TMPDIR=/tmp
PIN_IN=1 rocprofv3 --kernel-trace --memory-copy-trace -f csv -d $R/gpurun_out/wt_trace -o p -- python $R/tools/_wt.py > $R/gpurun_out/wt_trace.log 2>&1
cd $R
python tools/_trace_wt.py gpurun_out/wt_trace
