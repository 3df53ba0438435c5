"""Output side of the boundary: `OutputColorEncoder` and `FileOutput`.

Reference: `OutputColorEncoder::{to_output, from_output}` (`src/output/window.rs:105-115`,
`src/output/file.rs:61-70`) and `FileOutput::render_buffer` (`src/output/file.rs:27-49`).
The window output (minifb) is out of scope.
"""
from __future__ import annotations

import numpy as np

from .renderer import ImageBuffer


class WindowColorEncoder:
    """linear RGB f32 -> 0xFFRRGGBB: clamp to [0,1], x255, round half to even, no gamma."""

    @staticmethod
    def to_output(pixel) -> int:
        r, g, b = (float(c) for c in pixel)

        def u8(x: float) -> int:
            x = np.float32(x)
            if x != x:  # NaN -> 0, like fmaxf(NaN, 0)
                return 0
            x = np.float32(min(max(x, np.float32(0)), np.float32(1))) * np.float32(255)
            return int(np.rint(x))  # numpy rint = round half to even

        return 0xFF000000 | (u8(r) << 16) | (u8(g) << 8) | u8(b)

    @staticmethod
    def from_output(pixel: int):
        return (((pixel >> 16) & 0xFF) / 255.0, ((pixel >> 8) & 0xFF) / 255.0, (pixel & 0xFF) / 255.0)


FileColorEncoder = WindowColorEncoder  # the reference's two encoders are identical (file.rs:61-70)


class FileOutput:
    """`FileOutput::<W,H,_>::new(path).render_buffer(&buffer)`: u32 rows -> RGB8 PNG."""

    def __init__(self, path: str):
        self.path = path

    @staticmethod
    def new(path: str) -> "FileOutput":
        return FileOutput(path)

    def render_buffer(self, buffer: ImageBuffer) -> None:
        from PIL import Image

        Image.fromarray(buffer.as_rgb8()).save(self.path)
