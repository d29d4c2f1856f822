import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from enstop_amd.engine import Engine
e = Engine(0)
for mb in (2, 16, 24, 64, 128, 200, 512, 2048, 8192):
    print("read %5d MB : %8.1f GB/s" % (mb, e.stream_bandwidth(mb << 20, 3, 20 if mb < 1024 else 5)))
