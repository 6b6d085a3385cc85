cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/lbat; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_a -o run -- python3 $R/bench.py --steps 600 --warmup 50 --ctor-ahead 2 --no-cpu-baseline --no-secondary --no-dropin > $O/bench.log 2>&1
python3 $R/tools/lba_timeline.py $(find /tmp/tr_a -name "*kernel_trace.csv" | head -1) > $O/agent.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_b -o run -- python3 $R/tools/lba_time.py > $O/alone.log 2>&1
python3 $R/tools/lba_timeline.py $(find /tmp/tr_b -name "*kernel_trace.csv" | head -1) > $O/alone.txt 2>&1
