"""A real OpenEXR file to pin emfusion_amd.readers.read_exr: CPython ships the same 16 x 16 logo as
python.exr (written by the OpenEXR library: 4 HALF channels A, B, G, R, uncompressed) and python.png
in Lib/test/imghdrdata.  This copies the .exr (2.6 kB) and stores the PNG's pixels, decoded with
Pillow, as the expected values (uint8 RGBA) -- the two agree to half-float precision."""
import shutil
import sys
import sysconfig
from pathlib import Path

import numpy as np
from PIL import Image

HERE = Path(__file__).resolve().parent


def main():
    cands = [Path(sysconfig.get_paths()["stdlib"]) / "test" / "imghdrdata"] + \
        [p / "test" / "imghdrdata" for p in Path("/mnt/sandboxing").glob("**/lib/python3.*") if p.is_dir()]
    src = next(c for c in cands if (c / "python.exr").exists())
    shutil.copy(src / "python.exr", HERE / "python_logo.exr")
    rgba = np.asarray(Image.open(src / "python.png").convert("RGBA"))
    np.save(HERE / "python_logo_rgba.npy", rgba)
    print(src, rgba.shape)


if __name__ == "__main__":
    sys.exit(main())
