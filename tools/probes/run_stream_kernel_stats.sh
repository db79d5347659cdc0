set -e -o pipefail
R=$PWD; O=$R/gpurun_out/ks_stream4; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/s -- python3 $R/bench_streaming.py --seconds 6 --no-cpu-baseline > $O/bench.json 2> $O/err.txt
cd $R
python3 tools/per_grid_stats.py $(ls $O/s/*/*_kernel_trace.csv | head -1) 1 > $O/per_grid.txt
python3 tools/summarize_stats.py $(ls $O/s/*/*_kernel_stats.csv | head -1) 1 $O/stats.csv > /dev/null
rm -rf $O/s
