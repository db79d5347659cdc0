set -e -o pipefail
R=$PWD; O=$R/gpurun_out/dw1d; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for tt in 8 4; do
  export K2HIP_DW1D_TT=$tt
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/s$tt -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/bench$tt.json 2> $O/err$tt.txt
  python3 $R/tools/summarize_stats.py $(ls $O/s$tt/*/*_kernel_stats.csv | head -1) 22 $O/stats$tt.csv > /dev/null
done
