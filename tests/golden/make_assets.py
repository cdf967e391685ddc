"""Makes the data assets under raytracinginrust_amd/assets/ (teapot.obj, earthmap.jpg: shipped with the package) and the small decoder
fixtures under tests/golden/ from the reference's asset files (run once, in the build
container where /root/reference is mounted; the GPU box only sees the committed outputs).

  teapot.obj            byte copy of /root/reference/teapot.obj (Utah teapot mesh: 530 v, 1024 f) — input data
                        for BASELINE config 4 (tri.rs / mesh.rs path)
  earthmap.jpg          byte copy of /root/reference/earthmap.jpg — the texture BASELINE config 3 (final scene) and `earth` load
  earthmap_256x128.png  (small fixture for the JPEG-decoder tests only) /root/reference/earthmap.jpg (1024x512 baseline JPEG) decoded with Pillow and box-
                        downsampled 4x, stored losslessly — input texels for the ImageTexture path.  (The
                        reference decodes with the `image` crate; decoders may differ by +-1 LSB, SURVEY §8(c),
                        so the decoded texels, not the JPEG, are the fixture.)
"""
import os, shutil
from PIL import Image
here = os.path.dirname(os.path.abspath(__file__))
assets = os.path.join(os.path.dirname(os.path.dirname(here)), "raytracinginrust_amd", "assets")
os.makedirs(assets, exist_ok=True)
shutil.copyfile("/root/reference/teapot.obj", os.path.join(assets, "teapot.obj"))
# earthmap.jpg: byte copy of the reference's texture asset (1024x512 baseline 4:4:4 JPEG, loaded at src/main.rs:248,491-495) — the
# input data of BASELINE config 3; decoded at run time by the library's own JPEG ingest (csrc/rt_jpeg.cpp)
shutil.copyfile("/root/reference/earthmap.jpg", os.path.join(assets, "earthmap.jpg"))
im = Image.open("/root/reference/earthmap.jpg").convert("RGB")
assert im.size == (1024, 512)
im.resize((256, 128), Image.BOX).save(os.path.join(here, "earthmap_256x128.png"), optimize=True)
print("ok")


# A baseline 4:4:4 JPEG of the same texels (the reference's earthmap.jpg is baseline, 8-bit, 1x1-sampled YCbCr): input for
# the library's JPEG ingest test.  Encoded by Pillow from the fixture above; tests compare our decode with Pillow's.
Image.open(os.path.join(here, "earthmap_256x128.png")).save(os.path.join(here, "earthmap_256x128_444.jpg"), quality=90, subsampling=0, optimize=False, progressive=False)
Image.open(os.path.join(here, "earthmap_256x128.png")).convert("L").save(os.path.join(here, "earthmap_256x128_grey.jpg"), quality=85)
print("ok jpeg")
