cd /root/repo
python -m pytest tests -x -q -m gpu 2>&1 | tail -3
bash tools/profile_all.sh r05 C2 C1 C3 C4 C5 > gpurun_out/prof_r05.log 2>&1
tail -2 gpurun_out/prof_r05.log
