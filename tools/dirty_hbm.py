#!/usr/bin/env python3
"""Fill most of HBM with a byte pattern and free it: what hipMalloc hands to the next process is then no longer zero, so a
kernel that reads memory nobody wrote shows up in the tests run after it (a fresh box gives zeros and hides it).
usage: python3 tools/dirty_hbm.py [GiB, default 200] [byte, default 0xA5]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import genodsp_amd as gd  # noqa: E402
gib = int(sys.argv[1]) if len(sys.argv) > 1 else 200
byte = int(sys.argv[2], 0) if len(sys.argv) > 2 else 0xA5
gd.set_device(0)
bufs = []
for _ in range(gib // 8):
    b = gd.DeviceBuffer(8 << 30)
    gd.call("gdsp_memset", b.ptr, byte, 8 << 30, None)
    bufs.append(b)
gd.sync()
print("dirtied %d GiB with 0x%02X" % (8 * len(bufs), byte))
