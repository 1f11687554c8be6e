# usage (on the GPU box, via gpurun): bash profiles/microbench/kstats.sh <tag> <bench.py args...>
# rocprofv3 --kernel-trace --stats of one bench.py command; the kernel-stats CSV lands in gpurun_out/<tag>_kernel_stats.csv
tag=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks_$tag
rocprofv3 --kernel-trace --stats -d /tmp/ks_$tag -o run --output-format csv -- python $R/bench.py --no-secondary "$@" > $R/gpurun_out/${tag}_bench.json 2> /tmp/ks_$tag.err
f=$(find /tmp/ks_$tag -name "*kernel_stats.csv" | head -1)
cp "$f" $R/gpurun_out/${tag}_kernel_stats.csv
cut -d, -f1-4,8 $R/gpurun_out/${tag}_kernel_stats.csv | head -8
