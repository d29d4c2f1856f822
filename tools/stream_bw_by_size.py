import sys, json
sys.path.insert(0, '.')
from enstop_amd.engine import Engine
e = Engine(0)
out = {}
for mb in (64, 236, 512, 1024, 4096):
    out[mb] = {kind: round(e.stream_bandwidth(nbytes=mb << 20, kind=kind, reps=20), 1) for kind in (0, 1, 2)}
print(json.dumps({"stream_bandwidth_GBps_by_MB_and_kind": out}))
