#!/usr/bin/env python3
"""Box-downsamples the reference's only result artefact, /root/reference/output.png (1140x950,
semesterbild at the default feature set), by 4x4 into tests/golden/reference_output_box4.png.

The reference render is stochastic (unseeded AA jitter and light clouds), so this is a
STATISTICAL oracle: tests compare the 4x4-box-filtered GPU render against it (mean abs error,
silhouette IoU), never per pixel.  Run here once: python make_reference_output_fixture.py
"""
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
a = np.asarray(Image.open("/root/reference/output.png").convert("RGB")).astype(np.float32)
h, w, _ = a.shape
h4, w4 = h // 4 * 4, w // 4 * 4
b = a[:h4, :w4].reshape(h4 // 4, 4, w4 // 4, 4, 3).mean(axis=(1, 3))
Image.fromarray(np.clip(np.rint(b), 0, 255).astype(np.uint8)).save(os.path.join(HERE, "reference_output_box4.png"))
print(b.shape)
