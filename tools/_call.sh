mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr1 -o t -- python /root/repo/tools/one_utterance_trace.py > /dev/null 2>&1
cd /root/repo
f=$(find /tmp/tr1 -name "*kernel_trace.csv" | head -1)
python tools/timeline.py $f 60 > gpurun_out/one_utt_timeline.txt 2>&1
tail -70 gpurun_out/one_utt_timeline.txt
