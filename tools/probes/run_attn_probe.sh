# usage: run_attn_probe.sh <outdir> <probe values...>
R=$PWD; O=$R/gpurun_out/$1; shift; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for p in "$@"; do
  export K2HIP_ATTN_PROBE=$p
  rm -rf $O/trace
  rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-host-leg --no-secondary > $O/bench_$p.json 2> $O/bench_$p.err
  echo "probe $p rc $?" >> $O/summary.txt
  python3 $R/tools/per_grid_stats.py $(ls $O/trace/*/*_kernel_trace.csv | head -1) 9 k_attn >> $O/summary.txt
done
rm -rf $O/trace
cat $O/summary.txt
