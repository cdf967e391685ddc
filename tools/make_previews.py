"""Renders small previews of the four BASELINE scenes on the GPU into docs/previews/ (eyeball sanity next to the
reference's gallery images in img/; not a test)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from PIL import Image
from raytracinginrust_amd import _lib, render as R, scenes
be = _lib.load()
earth = scenes.load_earthmap()
out_dir = os.path.join(ROOT, 'gpurun_out', 'previews'); os.makedirs(out_dir, exist_ok=True)
jobs = [('cornell_box', scenes.cornell_box(be), 256, 256, 1024, 50), ('random_scene', scenes.random_scene(be, aspect_ratio=16 / 9), 384, 216, 256, 8),
        ('final_scene', scenes.final_scene(be, *earth), 256, 256, 1024, 50), ('cornell_test_teapot', scenes.cornell_test(be, scenes.asset_path('teapot.obj'), aspect_ratio=16 / 9), 384, 216, 512, 50)]
for name, (b, cam, bg), W, H, spp, depth in jobs:
    img = R.format_image(R.render(b, cam, bg, W, H, spp, depth), spp).astype(np.uint8)
    Image.fromarray(img).save(os.path.join(out_dir, name + '.png'), optimize=True)
    print(name, W, H, spp, f'{R.last_kernel_ms(b):.1f} ms')
