cd /tmp && export TMPDIR=/tmp
O=/root/repo/gpurun_out/prof_r05_default; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 /root/repo/bench.py > $O/bench.json 2> $O/bench.err
tail -c 200 $O/bench.json
cd /root/repo
RT_MULTI_VIRTUAL_RANKS=8 python bench.py --gpus 8 --steps 3 --warmup 1 --cpu-spp 0 > gpurun_out/bench_inproc8.json 2> gpurun_out/bench_inproc8.err; tail -c 200 gpurun_out/bench_inproc8.json
