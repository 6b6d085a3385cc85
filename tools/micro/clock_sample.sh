rocm-smi --showclocks 2>&1 | grep -i "sclk\|mclk\|fclk" | head -12
echo "--- during bench"
python3 bench.py --no-dropin --no-cpu-baseline --no-secondary --steps 60000 --warmup 100 > gpurun_out/clk_bench.json 2>/dev/null &
BP=$!
sleep 25
for i in 1 2 3 4 5 6; do rocm-smi --showclocks 2>&1 | grep -i "sclk" | head -8 | tr '\n' ' '; echo; sleep 0.5; done
rocm-smi --showpower --showperflevel 2>&1 | grep -i "power\|perf" | head -6
wait $BP
echo "--- during lba alone"
python3 tools/lba_time.py C2 20000 > gpurun_out/clk_lba.txt 2>&1 &
BP=$!
sleep 12
for i in 1 2 3 4; do rocm-smi --showclocks 2>&1 | grep -i "sclk" | head -8 | tr '\n' ' '; echo; sleep 0.5; done
wait $BP
tail -1 gpurun_out/clk_lba.txt
