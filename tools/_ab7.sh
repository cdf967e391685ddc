cd /root/repo
L=raytracinginrust_amd/csrc
for sc in cornell smoke final random; do python tools/ab_samples.py --scene $sc --size 128 --spp 16 r5a=$L/abx/r5a.so new=$L/librt_amd.so 2>&1 | grep -v amdgpu.ids; done
python tools/ab.py --scene cornell --spp 256 --rounds 5 r5a=$L/abx/r5a.so new=$L/librt_amd.so 2>&1 | grep -v amdgpu.ids
python tools/ab.py --scene final --spp 64 --rounds 6 r5a=$L/abx/r5a.so new=$L/librt_amd.so 2>&1 | grep -v amdgpu.ids
