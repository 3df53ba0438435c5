#!/usr/bin/env python3
"""Packs the reference's mesh DATA files (vertex / normal / index arrays and the MTL values the
loader reads) into hslu_i/ba_raytracing/f2501_raytracer_amd/data/*.npz.

The reference tree (/root/reference) does not exist on the GPU box, and the semesterbild scene of
BASELINE.json needs its text mesh as input.  Run here, once:

    python tools/pack_obj.py /root/reference/data/obj/text
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hslu_i.ba_raytracing.f2501_raytracer_amd.obj import pack_obj  # noqa: E402

src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/data/obj/text"
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                   "hslu_i", "ba_raytracing", "f2501_raytracer_amd", "data")
os.makedirs(dst, exist_ok=True)
for name in ("text", "text_lowres"):
    pack_obj(os.path.join(src, name + ".obj"), os.path.join(dst, name + ".npz"))
    print("packed", name, os.path.getsize(os.path.join(dst, name + ".npz")), "bytes")
