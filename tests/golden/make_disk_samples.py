#!/usr/bin/env python3
"""Samples the reference's shipped disk texture (/root/reference/src/renderer/textures/disk.png, the output of its
asset tool perlin/src/main.rs) at 4096 hashed positions and stores positions + values (derived data, 20 KB) so that the
restated generator stays pinned to a reference-produced artefact on machines without /root/reference."""
import os
import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
img = np.array(Image.open("/root/reference/src/renderer/textures/disk.png"))
assert img.shape == (1000, 1000, 4)
i = np.arange(4096, dtype=np.uint64)
x = ((i * np.uint64(2654435761)) >> np.uint64(7)) % np.uint64(1000)
y = ((i * np.uint64(40503) + np.uint64(977)) * np.uint64(2246822519) >> np.uint64(9)) % np.uint64(1000)
np.savez_compressed(os.path.join(HERE, "disk_png_samples.npz"), x=x.astype(np.int32), y=y.astype(np.int32),
                    rgba=img[y.astype(int), x.astype(int)], mean=np.array([img.mean()]), hist=np.bincount(img[..., 0].ravel(), minlength=256))
print("ok", img.mean())
