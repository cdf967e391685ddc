cd /root/repo
L=raytracinginrust_amd/csrc
V="new=$L/librt_amd.so bvh3=$L/abx/bvh3.so"
python tools/ab.py --scene random --spp 256 --rounds 4 $V 2>&1 | grep -v amdgpu.ids
python tools/ab.py --scene final --spp 64 --rounds 8 $V 2>&1 | grep -v amdgpu.ids
