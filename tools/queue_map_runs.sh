cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6; do
  rm -rf /tmp/kt$i
  rocprofv3 --kernel-trace --output-format csv -d /tmp/kt$i -o run -- python3 $R/bench.py --steps 400 --warmup 20 --no-cpu-baseline --no-secondary --no-dropin 2>/dev/null | grep "^{" | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('run $i', d['value'], d['step_ms_p50'], d['step_ms_p95'], d['config']['lba_ms_per_call'])"
  python3 $R/tools/queue_map.py $(find /tmp/kt$i -name "*kernel_trace.csv" | head -1)
done
