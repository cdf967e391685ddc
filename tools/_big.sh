cd /root/repo
python tools/mesh_size_probe.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_mesh_size_probe.log
cat gpurun_out/r05_mesh_size_probe.log
for k in 0 1 2 3 4; do
  (python tests/sweeps/list_scene_sweep.py $((1000000 + k * 12000)) 12000 > gpurun_out/r05_big_l$k.log 2>&1; python tests/sweeps/fuzz_sweep.py $((2000000 + k * 20000)) 20000 4000 > gpurun_out/r05_big_f$k.log 2>&1) &
done
wait
cat gpurun_out/r05_big_l*.log gpurun_out/r05_big_f*.log | grep -v amdgpu.ids
