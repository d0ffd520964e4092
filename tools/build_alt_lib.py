#!/usr/bin/env python3
"""A second build of the library for same-box A/Bs (PV_HIP_LIB=<out> selects it): ONE source recompiled with extra flags, linked with the in-tree objects.
usage: python tools/build_alt_lib.py <out.so> <source.hip> [flags ...]     (flags starting with '--extra=' replace nothing, they are appended)"""
import os
import subprocess
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import photoverse_amd.build as b  # noqa: E402
out, src, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
o = out + ".o"
subprocess.check_call([b._hipcc(), *b.FLAGS, *b.EXTRA_FLAGS.get(src, []), *flags, "-c", os.path.join(b.CSRC, src), "-o", o], stderr=subprocess.DEVNULL)
objs = [o if f == src else os.path.join(b.LIBDIR, f.replace(".hip", ".o")) for f in b.SOURCES]
subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs])
print("built", out)
