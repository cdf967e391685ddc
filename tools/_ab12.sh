cd /root/repo
L=raytracinginrust_amd/csrc
V="new=$L/librt_amd.so apre=$L/abx/apre.so"
for sc in random final; do python tools/ab_samples.py --scene $sc --size 128 --spp 16 base=$L/abx/base.so apre=$L/abx/apre.so 2>&1 | grep -v amdgpu.ids | tail -1; done
python tools/ab.py --scene random --spp 256 --rounds 5 $V 2>&1 | grep -v amdgpu.ids
python tools/ab.py --scene final --spp 64 --rounds 8 $V 2>&1 | grep -v amdgpu.ids
