set -e
R=$PWD
O=$R/gpurun_out/overlap
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-leg --no-secondary > $O/bench.json 2> $O/err.txt
cd $R
python3 tools/overlap_trace.py $(ls $O/trace/*/*_kernel_trace.csv | head -1) > $O/overlap.txt
cp $(ls $O/trace/*/*_kernel_trace.csv | head -1) $O/kernel_trace.csv; rm -rf $O/trace
tail -30 $O/overlap.txt
